#!/usr/bin/env python3
"""In-kernel anatomy of k_gemm9_tn: wall-clock stamps (100 MHz) of wave 0 of every workgroup at each tile's start (0), after its
first (1) and second (2) K-tile, at the end of its main loop (3), after the epilogue's stores are issued (4), and at the
workgroup's end (5, all stores acknowledged)."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from veritasfi_amd import _ffi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="51200x768x768,51200x2304x768,51200x768x3072")
    ap.add_argument("--epi", type=int, default=0)
    a = ap.parse_args()
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    L.vf_debug_gemm9_stamps.argtypes = [ctypes.c_void_p]
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
        for _ in range(3):
            L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, a.epi, st, 10)
        torch.cuda.synchronize()
        L.vf_debug_gemm9_stamps(buf.data_ptr())
        L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, a.epi, st, 10)
        torch.cuda.synchronize()
        L.vf_debug_gemm9_stamps(None)
        s = buf.cpu().numpy().astype(np.float64) / 100.0     # us
        t0 = s[:, 0, 0][s[:, 0, 0] > 0].min()
        nk = K // 64
        ntiles = (M // 256) * (N // 256)
        out = {"shape": sh, "stagger": os.environ.get("VF_GEMM_9_STAGGER", "100"), "tiles": ntiles, "nk": nk}
        starts = s[:, 0, 0] - t0
        ends = s[:, 15, 5] - t0
        out["wg_start_us"] = [round(float(np.percentile(starts, q)), 1) for q in (0, 50, 100)]
        out["wg_end_us"] = [round(float(np.percentile(ends, q)), 1) for q in (0, 50, 100)]
        per = []
        for it in range(min(16, -(-ntiles // 256))):
            ok = s[:, it, 3] > 0
            if not ok.any():
                break
            d = s[ok, it]
            row = {"tile": it, "wgs": int(ok.sum()),
                   "kt0": round(float(np.median(d[:, 1] - d[:, 0])), 2), "kt1": round(float(np.median(d[:, 2] - d[:, 1])), 2),
                   "rest_per_kt": round(float(np.median((d[:, 3] - d[:, 2]) / max(1, nk - 2))), 3),
                   "loop": round(float(np.median(d[:, 3] - d[:, 0])), 2)}
            if it > 0:
                prev = s[ok, it - 1, 3]
                row["gap_after_prev"] = round(float(np.median(d[:, 0] - prev)), 2)
            per.append(row)
        out["per_tile_median_us"] = per
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
