#!/usr/bin/env python3
"""GEMM micro-bench / parity through the vf_debug_gemm test hook: C = A[M,K] . W[N,K]^T + bias (fp16 in, fp32 acc).
Env: VF_GEMM_RING=1 selects the ring kernel, VF_GEMM_ABLATE bit 1 = no MFMA, 2 = no loads, 4 = no epilogue."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from veritasfi_amd import _ffi

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="51200x2304x768,51200x768x768,51200x3072x768,51200x768x3072")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--kind", type=int, default=0, help="0 auto, 5 DMA 128x256 (16x16x32 MFMA), 1 DMA 128x256 (32x32x16), 2 256x256, 3 128x128")
    a = ap.parse_args()
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    for sh in a.shapes.split(","):
        M, N, K = map(int, sh.split("x"))
        g = torch.Generator(device=dev).manual_seed(1)
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
        bias = torch.randn(N, device=dev, generator=g)
        C = torch.empty(M, N, device=dev, dtype=torch.float16)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, C.data_ptr(), M, N, K, 0, st, a.kind)
            assert rc == 0
        run(); torch.cuda.synchronize()
        err = None
        if a.check and not os.environ.get("VF_GEMM_ABLATE"):
            ref = (A[:4096].float() @ W.float().T + bias).half()
            err = float((C[:4096].float() - ref.float()).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        lib_us = None
        if a.check:
            lin = torch.nn.functional.linear
            bh = bias.half()
            for _ in range(3): lin(A, W, bh)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters): lin(A, W, bh)
            e1.record(); torch.cuda.synchronize()
            lib_us = round(e0.elapsed_time(e1) * 1e3 / a.iters, 1)
        print(json.dumps({"shape": sh, "us": round(us, 1), "vendor_lib_us": lib_us, "tflops": round(2 * M * N * K / us / 1e6, 1), "max_err": err,
                          "ring": os.environ.get("VF_GEMM_RING"), "ablate": os.environ.get("VF_GEMM_ABLATE")}), flush=True)

if __name__ == "__main__":
    main()
