#!/usr/bin/env python3
"""Fuzz of the decoder forward (HipDecoder, last-token pooling: step3_mul.py:181-209) against the HF Qwen3 module in torch fp32 on the
CPU: random batch sizes, padded lengths (1..700), padding side, length distributions; two geometries (grouped-query head dim 64
and 128).  Tolerances of tests/test_gpu_encoder.py.   python tools/fuzz_decoder.py --seconds 120 --seed 1"""
import argparse, importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import veritasfi_amd as vf
    from veritasfi_amd.retrieval import last_token_pool
    spec = importlib.util.spec_from_file_location("tge", os.path.join(ROOT, "tests", "test_gpu_encoder.py"))
    tge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tge)
    models = {"gqa-dh64": tge._hf_qwen3(256, 2, 4, 2, 64, 512), "gqa-dh128": tge._hf_qwen3(256, 2, 4, 1, 128, 512, seed=5)}
    decs = {k: vf.HipDecoder.from_hf(m, pooling=2, normalize=False) for k, m in models.items()}
    rng = np.random.default_rng(a.seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    t0 = time.time()
    n = fails = 0
    worst = [0.0, 0.0]
    try:
        while time.time() - t0 < a.seconds:
            kind = pick(list(models))
            b = pick([1, 1, 2, 3, 4, 8, 12, 16, int(rng.integers(1, 17))])
            t = pick([1, 2, 31, 32, 33, 64, 65, 100, 128, 129, 255, 256, 300, 511, 512, 513, 700, int(rng.integers(1, 701))])
            left = bool(rng.integers(2))
            dist = pick(["full", "tiny", "one_long", "random", "random"])
            if dist == "full":
                lens = np.full(b, t)
            elif dist == "tiny":
                lens = rng.integers(1, min(t, 4) + 1, size=b)
            elif dist == "one_long":
                lens = rng.integers(1, max(2, t // 8) + 1, size=b); lens[int(rng.integers(b))] = t
            else:
                lens = rng.integers(1, t + 1, size=b)
            lens = np.minimum(lens, t).astype(np.int64)
            lens[int(rng.integers(b))] = t          # a padded batch always has one full row (what a tokenizer produces)
            ids = rng.integers(5, 800, size=(b, t)).astype(np.int64)
            ar = np.arange(t)[None, :]
            mask = ((ar >= t - lens[:, None]) if left else (ar < lens[:, None])).astype(np.int64)
            with torch.no_grad():
                hs = models[kind](input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
                want = last_token_pool(hs, torch.from_numpy(mask)).numpy()
            got = decs[kind].forward(ids, mask)
            if np.isfinite(got).all():
                omc, rel = tge._embedding_errors(got, want)
            else:
                omc, rel = float("inf"), float("inf")
            worst = [max(worst[0], omc / tge.DEC_COS_TOL), max(worst[1], rel / tge.DEC_REL_TOL)]
            n += 1
            if not (omc < tge.DEC_COS_TOL and rel < tge.DEC_REL_TOL):
                fails += 1
                print("FAIL", json.dumps({"kind": kind, "b": b, "t": t, "left": left, "dist": dist, "lens": lens.tolist(), "one_minus_cos": omc, "rel": rel}), flush=True)
            if n % 50 == 0:
                print(f"... {n} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    finally:
        for d in decs.values():
            d.close()
    print(json.dumps({"cases": n, "failures": fails, "worst_over_tol": {"one_minus_cos": round(worst[0], 3), "rel": round(worst[1], 3)},
                      "seconds": round(time.time() - t0, 1), "seed": a.seed}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
