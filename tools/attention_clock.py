#!/usr/bin/env python3
"""In-kernel clock and cycles per key tile of k_attention2's loop (diagnostic build, vf_debug_attention kind 26)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from veritasfi_amd import _ffi
from tools.bench_attention import make_case, run
B, T, heads = 100, 512, 12
L = _ffi.lib()
L.vf_debug_attention.restype = ctypes.c_int
L.vf_debug_attention.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
qkv, mask = make_case(B, T, heads, dev)
H = heads * 64
ctx = torch.zeros(B * T * H + 256 * 8 * 8 * 4, dtype=torch.float16, device=dev)   # + per-wave stamp records
for _ in range(200):           # load the chip before the stamped launch
    run(L, 2, qkv, mask, B, T, heads, ctx)
run(L, 26, qkv, mask, B, T, heads, ctx)
torch.cuda.synchronize()
raw = ctx[B * T * H:].cpu().numpy().view(np.uint64).reshape(-1, 8)
raw = raw[raw[:, 5] > 0]
rec = raw.astype(np.float64)
st, en = raw[:, 6].astype(np.int64), raw[:, 7].astype(np.int64)
t0 = st.min()
print(f"kernel-wide (100 MHz clock): first start .. last end {(en.max() - t0)/100:.1f} us;  starts spread {(st.max() - t0)/100:.1f} us;  "
      f"ends: p10 {(np.percentile(en,10)-t0)/100:.1f} p50 {(np.percentile(en,50)-t0)/100:.1f} p90 {(np.percentile(en,90)-t0)/100:.1f} max {(en.max()-t0)/100:.1f} us")
for npw in sorted(set(raw[:, 5].tolist())):
    sel = raw[:, 5] == npw
    print(f"  workgroups with {npw} pairs: {sel.sum() // (T // 64)}  end p50 {(np.percentile(en[sel],50)-t0)/100:.1f} us  max {(en[sel].max()-t0)/100:.1f} us")
npairs = rec[:, 5]
clk = rec[:, 3] / rec[:, 4] * 100e6
print(f"waves {len(rec)}  in-kernel clock {np.median(clk)/1e9:.3f} GHz   kernel span per wave {np.median(rec[:,4])/100:.1f} us  pairs per workgroup {npairs.min():.0f}-{npairs.max():.0f}")
for name, col in (("  of the tail: closing barrier", 0), ("tile loops", 1), ("tail (O, next Q)", 2)):
    per = rec[:, col] / npairs
    print(f"{name:28s} per pair: median {np.median(per):8.0f} cycles = {np.median(per)/np.median(clk)*1e6:6.2f} us   p10 {np.percentile(per,10):8.0f} p90 {np.percentile(per,90):8.0f}")
print(f"cycles per 32-key tile (64 queries): {np.median(rec[:,1]/npairs)/(T//32):.0f}")

# which waves are slow?  records are laid out [workgroup][wave]
nw = T // 64
full = ctx[B * T * H:].cpu().numpy().view(np.uint64).reshape(-1, nw, 8)
ok = full[:, 0, 5] > 0
loops = full[ok][:, :, 1].astype(np.float64) / full[ok][:, :, 5].astype(np.float64)
print("loop cycles per pair by wave id (median over workgroups):", " ".join(f"{np.median(loops[:, w]):.0f}" for w in range(nw)))
print(f"within a workgroup: (max - min) / mean  median {np.median((loops.max(1) - loops.min(1)) / loops.mean(1)):.3f};  "
      f"workgroup means: p10 {np.percentile(loops.mean(1), 10):.0f} p50 {np.percentile(loops.mean(1), 50):.0f} p90 {np.percentile(loops.mean(1), 90):.0f}")
