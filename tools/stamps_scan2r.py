#!/usr/bin/env python3
"""Where a wave of k_scan2r spends a tile (debug bits 7 + 9: entry / image staged / first 64 tile starts / stream end / flushed on the
100-MHz constant clock): ISOLATED launches (ordered scans, whole chip), fp16 or e4m3 rows, any option / debug experiment bit.
usage: stamps_scan2r.py ROWS D f16|fp8 [opt=value ...]      (VF_DBG_EXTRA = experiment bits of the test variant)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard


def main():
    rows, d, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda", 0)
    corpus = make_shard(torch, 0, rows, d, dev, dt)
    g = torch.Generator(device=dev); g.manual_seed(4321)
    qs = [torch.randn((64, d), generator=g, device=dev) for _ in range(4)]
    out = (torch.empty((64, 100), dtype=torch.int64, device=dev), torch.empty((64, 100), dtype=torch.float32, device=dev))
    ix = vf.DenseIndex.from_e4m3(corpus.view(torch.uint8)) if dt == "fp8" else vf.DenseIndex(corpus)
    ix.set_option("scan_impl", 5)
    for o in sys.argv[4:]:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    ix.set_option("debug", 128 + 512 + int(os.environ.get("VF_DBG_EXTRA", "0")))
    for i in range(9):
        ix.search_begin(0, qs[i % 4], 100, out[0], out[1])
        ix.search_end(0)
    torch.cuda.synchronize()
    st = ix.stats()
    buf = np.zeros(1024 * 72, dtype=np.uint64)
    n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
    t = buf[:n].reshape(-1, 72).astype(np.int64)
    t = t[t[:, 0] > 0]
    ix.close()
    us = lambda x: x / 100.0
    print(f"{rows} x {d} {dt} {' '.join(sys.argv[4:])} debug+{os.environ.get('VF_DBG_EXTRA', '0')}: kernel {st.get('scan_kernel')}, {len(t)} waves, candidates {st.get('candidates')}, reruns {st.get('exact_reruns')}")
    if int(os.environ.get("VF_DBG_EXTRA", "0")) & 4096:      # cycle accounting (test variant): dbg[4..9] = wait, use, fill, head, epilogue cycles, segments
        total = t[:, 71].astype(float)
        names = ("waiting for a segment (vmcnt)", "LDS reads + matrix instructions", "LDS-DMA refills (4 per segment)", "tile head (claim, addresses)", "epilogue (filter, candidates)")
        segs, ntile = t[:, 9].astype(float), t[:, 69].astype(float)
        acc = 0.0
        for k_, nm in enumerate(names):
            v = t[:, 4 + k_].astype(float)
            per = v / (segs if k_ < 3 else ntile)
            acc = acc + v
            print(f"  {nm:36s} {100 * np.median(v / total):5.1f} % of the wave's cycles   median {np.median(per):7.0f} cycles per {'segment' if k_ < 3 else 'tile'}")
        print(f"  {'(unaccounted: start, flush, stamps)':36s} {100 * np.median(1 - acc / total):5.1f} %    segments per wave {np.median(segs):.0f}, tiles {np.median(ntile):.0f}, cycles {np.median(total):.0f}")
        t[:, 4:68] = 0
    tiles = (t[:, 4:68] - t[:, 68:69]) / 100.0
    tiles[t[:, 4:68] == 0] = np.nan
    dd = np.diff(tiles, axis=1)
    for x, y in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 63)):
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
    q("launch span (first entry -> flushed)", us(t[:, 3] - t[:, 68].min()))
    life = (t[:, 3] - t[:, 68]).astype(float)
    ok = (life > 0) & (t[:, 71] > 0)
    if ok.any():
        ghz = t[ok, 71] / life[ok] * 0.1          # shader cycles per 10-ns tick
        print(f"  in-kernel clock (s_memtime over the wave's life / 100-MHz clock)  median {np.median(ghz):.3f} GHz  p10 {np.percentile(ghz, 10):.3f}  p90 {np.percentile(ghz, 90):.3f}")


if __name__ == "__main__":
    main()
