#!/usr/bin/env python3
"""Where a K-slice of the persistent product kernel's whole-product cut spends its time (k_gemm9_tn<EPI, true>): wave 0's wall-clock
stamps at the slice's start (0), after its first (1) and second (2) K-tile, at the end of its main loop (3), behind the co-operative
finish + the epilogue of its own blocks (4: stores issued) and at the workgroup's end (5: stores acknowledged)."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from veritasfi_amd import _ffi

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="6656x768x3072,4096x768x3072,6656x1024x4096")
ap.add_argument("--epi", type=int, default=2)
a = ap.parse_args()
L = _ffi.lib()
L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
L.vf_debug_gemm9_stamps.argtypes = [ctypes.c_void_p]
L.vf_debug_gemm9_split_launches.restype = ctypes.c_longlong
dev = torch.device("cuda:0")
for sh in a.shapes.split(","):
    M, N, K = map(int, sh.split("x"))
    g = torch.Generator(device=dev).manual_seed(1)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    bias = torch.randn(N, device=dev, generator=g)
    R = torch.randn(M, N, device=dev, generator=g).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros((256, 16, 6), dtype=torch.int64, device=dev)
    run = lambda: L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, a.epi, st, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n0 = L.vf_debug_gemm9_split_launches()
    L.vf_debug_gemm9_stamps(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    L.vf_debug_gemm9_stamps(None)
    s = buf.cpu().numpy().astype(np.float64) / 100.0
    ok = s[:, 0, 3] > 0
    d = s[ok, 0]
    t0 = d[:, 0].min()
    end5 = s[ok, 15, 5]
    med = lambda x: round(float(np.median(x)), 2)
    print(json.dumps({"shape": sh, "epi": a.epi, "cut": bool(L.vf_debug_gemm9_split_launches() - n0), "slices": int(ok.sum()), "k_tiles_per_slice": None,
                      "start_spread_us": round(float(d[:, 0].max() - t0), 2), "first_k_tile_us": med(d[:, 1] - d[:, 0]), "second_k_tile_us": med(d[:, 2] - d[:, 1]),
                      "main_loop_us": med(d[:, 3] - d[:, 0]), "finish_and_epilogue_us": med(d[:, 4] - d[:, 3]),
                      "finish_p10_p90_us": [round(float(np.percentile(d[:, 4] - d[:, 3], q)), 2) for q in (10, 90)],
                      "store_drain_us": med(end5 - d[:, 4]), "workgroup_life_us": med(end5 - d[:, 0]), "launch_span_us": round(float(end5.max() - t0), 2)}), flush=True)
