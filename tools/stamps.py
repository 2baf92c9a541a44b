#!/usr/bin/env python3
"""Per-wave wall-clock stamps of the main scan (debug aid): where does a launch's time go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda", 0)
    corpus = make_shard(torch, 0, rows, 768, dev)
    g = torch.Generator(device=dev); g.manual_seed(4321)
    q = torch.randn((64, 768), generator=g, device=dev)
    ix = vf.DenseIndex(corpus)
    for o in sys.argv[3:]:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    for dbg in [128 + int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('0', '4', '2'))]:
        ix.set_option("debug", dbg)
        for _ in range(3):
            ix.search_device(q, 100)
        buf = np.zeros(2048 * 4, dtype=np.uint64)
        n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
        t = buf[:n].reshape(-1, 4).astype(np.int64)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0  # 100 MHz
        print(f"  kernel span (first wave start -> last flush end): {us[:, 3].max():.1f} us; rows/s per wave spread: "
              f"stream-end p50 - min {np.median(us[:, 1]) - us[:, 1].min():.1f}, max - p50 {us[:, 1].max() - np.median(us[:, 1]):.1f}")
        start, send, sync, fin = us[:, 0], us[:, 1], us[:, 2], us[:, 3]
        def st(x): return f"min {x.min():7.1f} p50 {np.median(x):7.1f} p90 {np.percentile(x,90):7.1f} p99 {np.percentile(x,99):7.1f} max {x.max():7.1f}"
        print(f"debug={dbg} waves={len(t)} st={ix.stats()['candidates']/64:.0f} cand/q")
        print("  wave start      ", st(start))
        print("  stream end      ", st(send))
        print("  stream duration ", st(send - start))
        print("  WG sync reached ", st(sync))
        print("  flush end       ", st(fin))
        print("  flush duration  ", st(fin - sync))
    ix.close()

if __name__ == "__main__":
    main()
