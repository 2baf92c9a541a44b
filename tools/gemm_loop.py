#!/usr/bin/env python3
"""One 51200 x 2304 x 768 fp16 product in a loop for --seconds (so that rocm-smi can be sampled beside it); prints the mean time.
--kind 10 = k_gemm9_tn through vf_debug_gemm, 0v = the vendor library (torch.nn.functional.linear)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veritasfi_amd import _ffi

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="10")
ap.add_argument("--data", default="random")
ap.add_argument("--seconds", type=float, default=8.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
M, N, K = 51200, 2304, 768
g = torch.Generator(device=dev).manual_seed(1)
A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
if a.data == "zeros":
    A.zero_(); W.zero_()
bias = torch.zeros(N, device=dev)
C = torch.empty(M, N, device=dev, dtype=torch.float16)
L = _ffi.lib()
L.vf_debug_gemm.restype = ctypes.c_int
L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
st = torch.cuda.current_stream().cuda_stream
bh = bias.half()
def once():
    if a.kind == "0v":
        torch.nn.functional.linear(A, W, bh)
    else:
        rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, C.data_ptr(), M, N, K, 0, st, int(a.kind))
        assert rc == 0, rc
for _ in range(5):
    once()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < a.seconds:
    for _ in range(200):
        once()
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(f"kind {a.kind} data {a.data}: {dt / n * 1e6:.1f} us per product, {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s over {dt:.1f} s")
