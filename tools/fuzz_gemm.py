#!/usr/bin/env python3
"""Fuzz of the product dispatch (vf_debug_gemm kind 0 = what the forwards call): random M / N / K / epilogue drawn around the gates of
gemm<EPI>() -- 128 tiles of 256 x 256, one and two rounds of the CUs, K = 256 / 2048, the 384-workgroup gate of the 128 x 256 kernel --
against torch fp32 on the same fp16 operands; every product is launched twice and must repeat bit for bit (the split-K tails sum
partials in a fixed order).   python tools/fuzz_gemm.py --seconds 120 --seed 1"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from veritasfi_amd import _ffi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    t0 = time.time()
    n = fails = 0
    worst = 0.0
    while time.time() - t0 < a.seconds:
        N = pick([128, 256, 384, 512, 768, 768, 1024, 1024, 1280, 2048, 2304, 2560, 3072, 4096, 128 * int(rng.integers(1, 40))])
        K = pick([64, 128, 192, 256, 320, 512, 768, 768, 1024, 1984, 2048, 2112, 3072, 4096, 64 * int(rng.integers(1, 70))])
        tiles_target = pick([1, 8, 77, 78, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 600, 800, 1200, int(rng.integers(1, 1300))])
        M = max(128, int(round(tiles_target * 65536 / N / 128)) * 128)
        if rng.random() < 0.5:
            M = (M + 255) // 256 * 256
        if float(M) * N * K > 6e11:
            M = max(128, int(6e11 / N / K) // 256 * 256)
        epi = pick([0, 1, 2, 11])
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 31)))
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
        bias = torch.randn(N, device=dev, generator=g)
        R = torch.randn(M, N, device=dev, generator=g).half()
        outs = []
        for _ in range(2):
            C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, epi,
                                 torch.cuda.current_stream().cuda_stream, 0)
            assert rc == 0, (rc, M, N, K, epi)
            outs.append(C)
        torch.cuda.synchronize()
        ref = A.float() @ W.float().T + bias
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 11:
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == 2:
            ref = ref.half().float() + R.float()
        err = (outs[0].float() - ref).abs().max().item()
        tol = 2e-2 * max(1.0, (K / 3072) ** 0.5)
        ok = (not torch.isnan(outs[0]).any().item()) and err < tol and torch.equal(outs[0], outs[1])
        worst = max(worst, err / tol)
        n += 1
        if not ok:
            fails += 1
            print("FAIL", json.dumps({"M": M, "N": N, "K": K, "epi": epi, "err": err, "tol": tol, "repeat": torch.equal(outs[0], outs[1])}), flush=True)
        del A, W, R, ref, outs, C
    print(json.dumps({"cases": n, "failures": fails, "worst_err_over_tol": round(worst, 3), "seconds": round(time.time() - t0, 1), "seed": a.seed}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
