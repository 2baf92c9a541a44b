#!/usr/bin/env python3
"""GEMM micro-bench / parity through the vf_debug_gemm test hook: C = A[M,K] . W[N,K]^T + bias (fp16 in, fp32 acc).
--kind takes a comma list and times every kernel on the same operands, interleaved with the vendor library
(torch.nn.functional.linear = hipBLASLt) on the same device."""
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
    ap.add_argument("--kind", default="0", help="comma list: 0 auto, 5 DMA 128x256 (16x16x32 MFMA), 1 DMA 128x256 (32x32x16), "
                                                "2 256x256, 3 128x128, 7 8-phase 256x256")
    ap.add_argument("--data", default="random", help="random | zeros | ones (operand bits toggle less: the MFMA draws less power, the clock stays up)")
    ap.add_argument("--epi", type=int, default=0, help="0 bias, 1 bias + GELU, 2 bias + residual")
    a = ap.parse_args()
    kinds = [int(x) for x in str(a.kind).split(",")]
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    sk_mode = None
    if os.environ.get("VF_SK_MODE") is not None:   # split-K of the 8-phase kernel: 0 off, 1 default policy, 2 every partial round
        L.vf_debug_splitk_tail.restype = ctypes.c_int
        L.vf_debug_splitk_tail.argtypes = [ctypes.c_int]
        sk_mode = int(os.environ["VF_SK_MODE"])
        L.vf_debug_splitk_tail(sk_mode)
    def sk_stats():
        if sk_mode is None:
            return None
        out = (ctypes.c_uint * 2)()
        L.vf_debug_splitk_stats.restype = ctypes.c_int
        L.vf_debug_splitk_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.vf_debug_splitk_stats(out, int(os.environ.get("VF_SK_DBG", "0")))
        return [int(out[0]), int(out[1])]
    sk_stats()
    dev = torch.device("cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for sh in a.shapes.split(","):
        M, N, K = map(int, sh.split("x"))
        g = torch.Generator(device=dev).manual_seed(1)
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
        if a.data == "zeros":
            A.zero_(); W.zero_()
        elif a.data == "ones":
            A.fill_(1.0); W.fill_(1.0 / 64)
        bias = torch.randn(N, device=dev, generator=g)
        R = torch.randn(M, N, device=dev, generator=g).half()
        C = torch.empty(M, N, device=dev, dtype=torch.float16)
        st = torch.cuda.current_stream().cuda_stream
        lib_us = None
        if a.check:
            lin = torch.nn.functional.linear
            bh = bias.half()
            for _ in range(3):
                lin(A, W, bh)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                lin(A, W, bh)
            e1.record()
            torch.cuda.synchronize()
            lib_us = round(e0.elapsed_time(e1) * 1e3 / a.iters, 1)
        for kind in kinds:
            def run():
                rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, a.epi, st, kind)
                assert rc == 0, rc
            run()
            torch.cuda.synchronize()
            err = None
            if a.check and a.epi == 0:
                rows = torch.cat([torch.arange(0, min(M, 2048), device=dev), torch.arange(max(0, M - 2048), M, device=dev)])   # head and tail
                ref = (A[rows].float() @ W.float().T + bias).half()
                err = float((C[rows].float() - ref.float()).abs().max())
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print(json.dumps({"shape": sh, "kind": kind, "epi": a.epi, "us": round(us, 1), "vendor_lib_us": lib_us,
                              "tflops": round(2 * M * N * K / us / 1e6, 1), "vs_vendor": None if lib_us is None else round(us / lib_us, 3),
                              "max_err": err, "sk_mode": sk_mode, "sk_readbacks_l2_mem": sk_stats()}), flush=True)


if __name__ == "__main__":
    main()
