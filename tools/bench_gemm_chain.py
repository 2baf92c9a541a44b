#!/usr/bin/env python3
"""The four products of an encoder layer run as a dependent chain (each reads what the previous one wrote, as in the forward),
vendor library (torch.nn.functional.linear) against this repo's kernels (vf_debug_gemm, automatic kernel choice), bias-only
epilogue on both sides.  Isolated back-to-back timings flatter a kernel whose operands stay hot; this is the in-situ figure."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from veritasfi_amd import _ffi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="6656,12800,25600,51200")
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--ffn", type=int, default=3072)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--kind", type=int, default=0, help="force a kernel of this repo (7 = 8-phase, 8 = persistent 8-phase: experiments build)")
    ap.add_argument("--forward-epilogues", action="store_true", help="this repo's side runs the forward's epilogues: bias | bias + residual | bias + GELU | bias + residual")
    a = ap.parse_args()
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    H, F = a.hidden, a.ffn
    g = torch.Generator(device=dev).manual_seed(3)
    W = {n: (torch.randn(o, i, device=dev, generator=g) * 0.03).half() for n, (o, i) in
         {"qkv": (3 * H, H), "o": (H, H), "up": (F, H), "down": (H, F)}.items()}
    B = {n: torch.randn(w.shape[0], device=dev, generator=g) * 0.1 for n, w in W.items()}
    Bh = {n: b.half() for n, b in B.items()}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream().cuda_stream
    for M in [int(x) for x in a.rows.split(",")]:
        x = (torch.randn(M, H, device=dev, generator=g) * 0.5).half()
        bufs = {"qkv": torch.empty(M, 3 * H, device=dev, dtype=torch.float16), "o": torch.empty(M, H, device=dev, dtype=torch.float16),
                "up": torch.empty(M, F, device=dev, dtype=torch.float16), "down": torch.empty(M, H, device=dev, dtype=torch.float16)}

        def vendor():
            torch.nn.functional.linear(x, W["qkv"], Bh["qkv"], out=None)   # (no out= for linear: allocation is cached by torch)
            y = torch.nn.functional.linear(x, W["o"], Bh["o"])
            u = torch.nn.functional.linear(y, W["up"], Bh["up"])
            return torch.nn.functional.linear(u, W["down"], Bh["down"])

        def mine():
            def mm(A, n, C):
                epi = {"qkv": 0, "o": 2, "up": 1, "down": 2}[n] if a.forward_epilogues else 0
                rc = L.vf_debug_gemm(A.data_ptr(), W[n].data_ptr(), B[n].data_ptr(), x.data_ptr() if epi == 2 else None, C.data_ptr(), A.shape[0], W[n].shape[0], A.shape[1], epi, st, a.kind)
                assert rc == 0, rc
            mm(x, "qkv", bufs["qkv"]); mm(x, "o", bufs["o"]); mm(bufs["o"], "up", bufs["up"]); mm(bufs["up"], "down", bufs["down"])
            return bufs["down"]
        res = {}
        for name, fn in (("vendor", vendor), ("repo", mine), ("vendor2", vendor), ("repo2", mine)):
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = round(e0.elapsed_time(e1) * 1e3 / a.iters, 1)
        err = float((vendor().float() - mine().float()).abs().max())
        flops = 2.0 * M * (3 * H * H + H * H + 2 * H * F)
        print(json.dumps({"kind": a.kind, "forward_epilogues": a.forward_epilogues, "rows": M, "hidden": H, "ffn": F, "chain_us": res, "repo_over_vendor": round(min(res["repo"], res["repo2"]) / min(res["vendor"], res["vendor2"]), 3),
                          "repo_tflops": round(flops / min(res["repo"], res["repo2"]) / 1e6, 1), "max_abs_diff": err}), flush=True)


if __name__ == "__main__":
    main()
