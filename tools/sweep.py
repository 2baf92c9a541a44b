#!/usr/bin/env python3
"""In-process A/B sweep of index options on the GPU box (tuning aid, not a test).
usage: python tools/sweep.py ROWS [ROWS...]   -> one table per corpus size"""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import veritasfi_amd as vf
from bench import make_shard

CONFIGS = [
    {},
    {"debug": 4}, {"debug": 2},
    {"refresh_every": 16}, {"refresh_every": 64},
    {"scan_g": 1}, {"scan_g": 3},
    {"sample_rows": 32}, {"margin": 16},
]
DEFAULTS = {"debug": 0, "scan_g": 0, "sample_rows": 16, "refresh_every": 32, "waves": 0, "margin": -1}

def run(ix, q, k, steps, outs):
    pend = []
    for i in range(steps):
        s = i % 2
        if len(pend) == 2:
            ix.search_end(pend.pop(0))
        ix.search_begin(s, q, k, outs[s][0], outs[s][1])
        pend.append(s)
    while pend:
        ix.search_end(pend.pop(0))
    torch.cuda.synchronize()

def main():
    dev = torch.device("cuda", 0)
    for rows in [int(x) for x in sys.argv[1:]] or [1_000_000]:
        corpus = make_shard(torch, 0, rows, 768, dev)
        g = torch.Generator(device=dev); g.manual_seed(4321)
        q = torch.randn((64, 768), generator=g, device=dev)
        ix = vf.DenseIndex(corpus)
        outs = [(torch.empty((64, 100), dtype=torch.int64, device=dev), torch.empty((64, 100), dtype=torch.float32, device=dev)) for _ in range(2)]
        steps = 40 if rows <= 2_000_000 else 12
        print(f"== rows={rows} steps={steps}")
        for rep in range(2):
            for cfg in CONFIGS:
                for kname, v in DEFAULTS.items():
                    ix.set_option(kname, cfg.get(kname, v))
                run(ix, q, 100, 4, outs)
                ix.set_option("profile", 1)
                t0 = time.perf_counter()
                run(ix, q, 100, steps, outs)
                dt = (time.perf_counter() - t0) / steps
                p = ix.profile(); st = ix.stats()
                scan_ms = p["scan_ms_total"] / max(1, p["scan_launches"])
                gbs = p["scan_bytes_per_launch"] / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0
                print(f"rep{rep} {json.dumps(cfg):28s} step {dt*1e3:7.3f} ms  scan {scan_ms:7.3f} ms {gbs:7.0f} GB/s  pipe {p['pipeline_ms_total']/max(1,p['scan_launches']):7.3f} ms  cand/q {st['candidates']/64:7.0f} reruns {st['exact_reruns']}", flush=True)
        ix.close(); del corpus
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
