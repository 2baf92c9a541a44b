#!/usr/bin/env python3
"""Where k_scan_wide8's waves spend a launch (debug bit 7): waiting for their own DMAs at the K-tile boundary, in the K-tile
barrier, in the epilogues.  usage: stamps_wide8.py ROWS [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    dev = torch.device("cuda", 0)
    corpus = make_shard(torch, 0, rows, 1024, dev, "fp8")
    g = torch.Generator(device=dev); g.manual_seed(4321)
    q = torch.randn((1024, 1024), generator=g, device=dev)
    ix = vf.DenseIndex(corpus)
    ix.set_option("wide_mfma", 1)
    for o in sys.argv[2:]:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    ix.set_option("debug", 128)
    for _ in range(3):
        ix.search_device(q, 1000)
    buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
    n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
    t = buf[:n].reshape(-1, 16).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    ktiles = (rows // 64 - 512 + 255) // 256 * 16
    print(f"rows {rows}: {len(t)} waves, ~{ktiles} K-tiles per workgroup")
    for name, c in (("kernel", 0), ("wait for own DMAs", 1), ("K-tile barrier", 2), ("epilogues", 3), (" of which 1/norm load", 4), (" filter + candidates", 5), ("  thresholds + pass 1a", 6), ("  pass 1b (values)", 7), ("  pass 2 (stage)", 8), ("  publish + refresh", 9)):
        v = t[:, c]
        print(f"  {name:20s} median {np.median(v):8.1f} us  p10 {np.percentile(v, 10):8.1f}  p90 {np.percentile(v, 90):8.1f}   per K-tile {np.median(v) / ktiles:6.3f} us")
    rest = t[:, 0] - t[:, 1] - t[:, 2] - t[:, 3]
    print(f"  {'rest (reads + MFMA issue)':20s} median {np.median(rest):8.1f} us   per K-tile {np.median(rest) / ktiles:6.3f} us")
    ix.close()


if __name__ == "__main__":
    main()
