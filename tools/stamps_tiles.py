#!/usr/bin/env python3
"""Per-tile start times of k_scan2's waves (debug bits 7 + 9): is a wave's time per tile uniform over the launch?
usage: stamps_tiles.py ROWS [opt=value ...]"""
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
    corpus = make_shard(torch, 0, rows, 768, dev)
    g = torch.Generator(device=dev); g.manual_seed(4321)
    q = torch.randn((64, 768), generator=g, device=dev)
    ix = vf.DenseIndex(corpus)
    for o in sys.argv[2:]:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    ix.set_option("debug", 128 + 512 + int(os.environ.get("VF_DBG_EXTRA", "0")))
    for _ in range(3):
        ix.search_device(q, 100)
    nw = 1024
    buf = np.zeros(nw * 72, dtype=np.uint64)
    n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
    t = buf[:n].reshape(-1, 72).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    tiles = (t[:, 4:68] - t0) / 100.0
    tiles[t[:, 4:68] == 0] = np.nan
    d = np.diff(tiles, axis=1)
    print(f"rows {rows}: {len(t)} waves; first tile starts at {np.nanmedian(tiles[:, 0]):.1f} us (median), stream ends {np.median((t[:, 1] - t0) / 100.0):.1f}")
    for a, b in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 63)):
        seg = d[:, a:b]
        if np.all(np.isnan(seg)):
            continue
        print(f"  tiles {a:2d}..{b:2d}: time per tile  median {np.nanmedian(seg):6.2f} us  p10 {np.nanpercentile(seg, 10):6.2f}  p90 {np.nanpercentile(seg, 90):6.2f}")
    ix.close()
    # the life of a workgroup (its four waves share the entry and the barriers): entry -> image staged -> stream end -> flushed
    ent = t[:, 68]
    e0 = ent.min()
    us = lambda x: x / 100.0
    def q(name, v):
        print(f"  {name:34s} median {np.median(v):7.1f}  p10 {np.percentile(v, 10):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f} us")
    q("entry after the first entry", us(ent - e0))
    q("entry -> image staged", us(t[:, 0] - ent))
    q("image staged -> stream end", us(t[:, 1] - t[:, 0]))
    q("stream end -> barrier", us(t[:, 2] - t[:, 1]))
    q("barrier -> flushed", us(t[:, 3] - t[:, 2]))
    q("first entry -> flushed", us(t[:, 3] - e0))
    q("tiles taken by a wave", t[:, 69].astype(float))


if __name__ == "__main__":
    main()
