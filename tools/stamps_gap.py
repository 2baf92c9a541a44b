#!/usr/bin/env python3
"""What sits between two consecutive main-scan workgroups on ONE compute unit in the pipelined loop (debug bits 7 + 9 of k_scan2:
per-wave entry / image staged / stream end / flushed on the 100-MHz constant clock, plus HW_ID and XCC_ID).
The two slots' stamp buffers hold the last two launches; a CU's workgroup of the later launch is paired with the one of the earlier.
usage: stamps_gap.py ROWS [D] [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard


def read(ix, slot, nw=1024):
    buf = np.zeros(nw * 72, dtype=np.uint64)
    n = _ffi.lib().vf_index_debug_read(ix._h, slot, buf.ctypes.data, buf.size)
    t = buf[:n].reshape(-1, 72).astype(np.int64)
    return t[t[:, 0] > 0]


def cu_key(w):
    hw, xcc = w & 0xFFFFFFFF, w >> 32
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx950: se 3 bits)
    return (xcc & 0xF) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    args = sys.argv[2:]
    d = 768
    if args and "=" not in args[0]:
        d = int(args.pop(0))
    dev = torch.device("cuda", 0)
    corpus = make_shard(torch, 0, rows, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(4321)
    qs = [torch.randn((64, d), generator=g, device=dev) for _ in range(4)]
    outs = [(torch.empty((64, 100), dtype=torch.int64, device=dev), torch.empty((64, 100), dtype=torch.float32, device=dev)) for _ in range(2)]
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(side)
    ix = vf.DenseIndex(corpus)
    for o in args:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    ix.set_option("debug", 128 + 512 + int(os.environ.get("VF_DBG_EXTRA", "0")))
    steps = 41   # odd: the last launch is slot 0's, the one before slot 1's
    pend = []
    for i in range(steps):
        s = i % 2
        if len(pend) == 2:
            ix.search_end(pend.pop(0))
        ix.search_begin(s, qs[i % 4], 100, outs[s][0], outs[s][1])
        pend.append(s)
    while pend:
        ix.search_end(pend.pop(0))
    torch.cuda.synchronize()
    late, early = read(ix, 0), read(ix, 1)
    ix.close()
    us = lambda x: x / 100.0
    for name, t in (("earlier launch", early), ("later launch", late)):
        e0 = t[:, 68].min()
        print(f"{name}: {len(t)} waves on {len(set(cu_key(w) for w in t[:, 70]))} CUs; entries spread p10 {np.percentile(us(t[:, 68] - e0), 10):.1f} p50 {np.median(us(t[:, 68] - e0)):.1f} "
              f"p90 {np.percentile(us(t[:, 68] - e0), 90):.1f} max {us(t[:, 68] - e0).max():.1f} us; life (entry -> flushed) median {np.median(us(t[:, 3] - t[:, 68])):.1f} "
              f"p10 {np.percentile(us(t[:, 3] - t[:, 68]), 10):.1f} p90 {np.percentile(us(t[:, 3] - t[:, 68]), 90):.1f}; launch span {us(t[:, 3].max() - e0):.1f} us")
    # the later launch taken apart like tools/stamps_tiles.py does an isolated one
    t = late
    tiles = (t[:, 4:68] - t[:, 68:69]) / 100.0
    tiles[t[:, 4:68] == 0] = np.nan
    dd = np.diff(tiles, axis=1)
    for x, y in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 63)):
        seg = dd[:, x:y]
        if np.all(np.isnan(seg)):
            continue
        print(f"  tiles {x:2d}..{y:2d}: time per tile  median {np.nanmedian(seg):6.2f} us  p10 {np.nanpercentile(seg, 10):6.2f}  p90 {np.nanpercentile(seg, 90):6.2f}")
    def q(name, v):
        print(f"  {name:34s} median {np.median(v):7.1f}  p10 {np.percentile(v, 10):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f} us")
    q("entry -> image staged", us(t[:, 0] - t[:, 68]))
    q("image staged -> stream end", us(t[:, 1] - t[:, 0]))
    q("stream end -> barrier", us(t[:, 2] - t[:, 1]))
    q("barrier -> flushed", us(t[:, 3] - t[:, 2]))
    q("tiles taken by a wave", t[:, 69].astype(float))
    # a CU's two workgroups: per CU the (one) workgroup of each launch = min entry / max flushed over its four waves
    def per_cu(t):
        out = {}
        for row in t:
            k = cu_key(row[70])
            ent, fl = row[68], row[3]
            if k in out:
                out[k] = (min(out[k][0], ent), max(out[k][1], fl))
            else:
                out[k] = (ent, fl)
        return out
    a, b = per_cu(early), per_cu(late)
    gaps = np.array([us(b[k][0] - a[k][1]) for k in b if k in a])
    print(f"CUs seen in both launches: {len(gaps)}")
    if len(gaps):
        print(f"gap flushed(earlier) -> entry(later) on the same CU: median {np.median(gaps):.1f} p10 {np.percentile(gaps, 10):.1f} p90 {np.percentile(gaps, 90):.1f} "
              f"min {gaps.min():.1f} max {gaps.max():.1f} us")
        period = np.array([us(b[k][0] - a[k][0]) for k in b if k in a])
        print(f"entry(earlier) -> entry(later) on the same CU (the CU's period): median {np.median(period):.1f} p10 {np.percentile(period, 10):.1f} p90 {np.percentile(period, 90):.1f} us")
    # dispatch order against CU release order: does the later launch start its workgroups in the order the CUs came free?
    e_sorted = np.sort(late[:, 68][::4]) if len(late) % 4 == 0 else np.sort(late[:, 68])
    f_sorted = np.sort(np.array([v[1] for v in a.values()]))
    m = min(len(e_sorted), len(f_sorted))
    lag = us(e_sorted[:m] - f_sorted[:m])
    print(f"i-th entry of the later launch minus i-th release of the earlier: median {np.median(lag):.1f} p10 {np.percentile(lag, 10):.1f} p90 {np.percentile(lag, 90):.1f} us")


if __name__ == "__main__":
    main()
