#!/usr/bin/env python3
"""Where and when the workgroups of k_scan_wide8 ran (stamps build, debug bit 7): per XCC / CU the workgroups it hosted, their start
and end times, and how many workgroups the chip ran at once over the launch.  usage: stamps_wide8_where.py ROWS waves [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard

rows, waves = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
corpus = make_shard(torch, 0, rows, 1024, dev, "fp8")
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((1024, 1024), generator=g, device=dev)
ix = vf.DenseIndex(corpus)
ix.set_option("wide_mfma", 1); ix.set_option("wide8_waves", waves)
for o in sys.argv[3:]:
    name, val = o.split("=")
    ix.set_option(name, int(val))
ix.set_option("debug", 128)
for _ in range(3):
    ix.search_device(q, 1000)
nwg = 256 * 8 // waves
buf = np.zeros(nwg * waves * 16, dtype=np.uint64)
n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
t = buf[:n].reshape(-1, waves, 16)
w0 = t[:, 0, :]                                   # wave 0 of every workgroup
ok = w0[:, 0] > 0
start = w0[ok, 10].astype(np.float64) / 100.0
dur = w0[ok, 0].astype(np.float64) / 100.0
hw = (w0[ok, 11] & 0xFFFFFFFF).astype(np.int64); xcc = (w0[ok, 11] >> 32).astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
t0 = start.min()
start -= t0
end = start + dur
print(f"waves {waves}: {ok.sum()} workgroups, launch spans {end.max():.0f} us; workgroup life median {np.median(dur):.0f} us, starts: "
      f"{(start < 50).sum()} within 50 us, {(start >= 50).sum()} later (median late start {np.median(start[start >= 50]) if (start >= 50).any() else 0:.0f} us)")
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print(f"  distinct (xcc, se, sh, cu) places: {len(u)}; workgroups per place: " + ", ".join(f"{c}: {int((cnt == c).sum())}" for c in sorted(set(cnt))))
for when in (100, 0.25 * end.max(), 0.5 * end.max(), 0.75 * end.max()):
    print(f"  workgroups running at t = {when:7.0f} us: {int(((start <= when) & (end > when)).sum())}")
ix.close()
