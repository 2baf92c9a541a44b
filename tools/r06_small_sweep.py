#!/usr/bin/env python3
"""Round 6: where a small shard's step goes, by CU split / sample size / pipeline depth (one process, one box: every
figure of a table is comparable with the others of that table and with nothing else).
usage: r06_small_sweep.py ROWS D [--quick]
Per configuration a FRESH index (aux_cus is fixed once the scan streams exist): step = wall time of the pipelined loop
(two batches in flight, as bench.py), launch = average main-scan duration by events on its own stream, span = launch
interval (makespan / launches)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import veritasfi_amd as vf
from bench import make_shard


def run(ix, q, k, steps, outs, depth):
    pend = []
    for i in range(steps):
        s = i % depth
        if len(pend) == depth:
            ix.search_end(pend.pop(0))
        ix.search_begin(s, q, k, outs[s][0], outs[s][1])
        pend.append(s)
    while pend:
        ix.search_end(pend.pop(0))
    torch.cuda.synchronize()


def main():
    rows, d = int(sys.argv[1]), int(sys.argv[2])
    quick = "--quick" in sys.argv
    dev = torch.device("cuda", 0)
    corpus = make_shard(torch, 0, rows, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(4321)
    q = torch.randn((64, d), generator=g, device=dev)
    outs = [(torch.empty((64, 100), dtype=torch.int64, device=dev), torch.empty((64, 100), dtype=torch.float32, device=dev)) for _ in range(4)]
    bytes_alg = rows * (2 * d + 4)
    cfgs = [{"aux_cus": 32, "sample_rows": 4}, {"aux_cus": 32, "sample_rows": 8}, {"aux_cus": 32, "sample_rows": 16}, {"aux_cus": 32, "sample_rows": 32},
            {"aux_cus": 32, "sample_rows": 16, "sample_grid": 64}, {"aux_cus": 32, "sample_rows": 16, "sample_grid": 224},
            {"aux_cus": 32, "sample_rows": 16, "debug": 4}, {"aux_cus": 32, "sample_rows": 16, "refresh_every": 64},
            {"aux_cus": 32, "sample_rows": 16, "refresh_every": 32},
            {"aux_cus": 0, "overlap_scans": 1, "sample_rows": 16}, {"aux_cus": 0, "overlap_scans": 1, "sample_rows": 4},
            {"aux_cus": 0, "overlap_scans": 0, "sample_rows": 16}, {"aux_cus": 0, "overlap_scans": 1, "sample_rows": 16, "debug": 4},
            {"aux_cus": 64, "sample_rows": 16},
            {"aux_cus": 32, "sample_rows": 16, "depth": 3}, {"aux_cus": 32, "sample_rows": 16, "depth": 4}]
    # (aux_cus that are not a multiple of 32 -- one CU per shader engine -- leave some engines with more workgroups than CUs: a scan
    #  then takes two rounds, 0.65 ms per launch; profiles/r06_small_sweep_null_stream_and_uneven_masks.log)
    if os.environ.get("R06_CFGS"):
        cfgs = json.loads(os.environ["R06_CFGS"])
    if quick:
        cfgs = cfgs[:3]
    steps = 300
    print(f"== rows={rows} d={d} steps={steps} algorithmic bytes per launch {bytes_alg / 1e9:.3f} GB", flush=True)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(side)
    for rep in range(int(os.environ.get('R06_REPS', '2'))):
        for cfg in cfgs:
            cfg = dict(cfg)
            depth = cfg.pop("depth", 2)
            ix = vf.DenseIndex(corpus)
            for name, v in cfg.items():
                ix.set_option(name, v)
            run(ix, q, 100, 20, outs, depth)
            ix.set_option("profile", 1)
            t0 = time.perf_counter()
            run(ix, q, 100, steps, outs, depth)
            dt = (time.perf_counter() - t0) / steps
            p = ix.profile(); st = ix.stats()
            sp = p["span_ms"] / max(1, p["scan_launches"])
            scan_ms = p["scan_ms_total"] / max(1, p["scan_launches"])
            print(f"rep{rep} {json.dumps(cfg):64s} depth {depth} step {dt * 1e3:7.4f} ms = {bytes_alg / dt / 8e12:6.4f} of peak | launch {scan_ms:7.4f} ms"
                  f" | interval {sp:7.4f} ms | cand/q {st['candidates'] / 64:7.0f} reruns {st['exact_reruns']}", flush=True)
            ix.close()


if __name__ == "__main__":
    main()
