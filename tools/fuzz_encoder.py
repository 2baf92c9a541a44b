#!/usr/bin/env python3
"""Fuzz of the encoder forward (HipEncoder: padded and packed paths) against the HF module in torch fp32 on the CPU: random batch
sizes (1..48), padded lengths (1..512) and length distributions (all full, all tiny, one long + many short, random, multiples of
32 and one off them), for a BERT-style embedder (CLS + L2) and an XLM-R-style cross-encoder (logit).  Same tolerances as
tests/test_gpu_encoder.py.   python tools/fuzz_encoder.py --seconds 120 --seed 1"""
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
    ap.add_argument("--big", action="store_true", help="real widths (768 / 12 heads / 3072 and 1024 / 16 heads / 4096, 2 layers), batches of up to 100 "
                                                       "x 512 tokens: the forwards then run the LARGE product kernels and their dispatch gates; the HF "
                                                       "side runs in torch fp32 on the GPU")
    a = ap.parse_args()
    import veritasfi_amd as vf
    spec = importlib.util.spec_from_file_location("tge", os.path.join(ROOT, "tests", "test_gpu_encoder.py"))
    tge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tge)
    models = {
        "bert-embedder": (tge._hf_bert(256, 3, 4, 512), 0, 8e-4),
        "bert-embedder-2heads": (tge._hf_bert(128, 2, 2, 256, seed=3), 0, 8e-4),
        "xlmr-reranker": (tge._hf_xlmr_cls(256, 3, 4, 512), 1, 2.5e-3),
    }
    dev = "cpu"
    if a.big:
        dev = "cuda"
        models = {
            "bert-embedder-768": (tge._hf_bert(768, 2, 12, 3072).to(dev), 0, 8e-4),
            "xlmr-reranker-1024": (tge._hf_xlmr_cls(1024, 2, 16, 4096).to(dev), 1, 2.5e-3),
        }
    encs = {k: vf.HipEncoder.from_hf(m[0]) for k, m in models.items()}
    rng = np.random.default_rng(a.seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    t0 = time.time()
    n = fails = 0
    worst = {}
    try:
        while time.time() - t0 < a.seconds:
            kind = pick(list(models))
            m, pad_id, tol = models[kind]
            b = pick([1, 1, 2, 3, 7, 8, 13, 16, 24, 25, 32, 33, 48, int(rng.integers(1, 49))])
            if a.big:
                b = pick([1, 7, 13, 16, 25, 26, 32, 50, 51, 64, 100, int(rng.integers(1, 101))])
            t = pick([1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 300, 384, 511, 512, int(rng.integers(1, 513))])
            dist = pick(["full", "tiny", "one_long", "random", "random", "mult32", "off32"])
            if dist == "full":
                lens = np.full(b, t)
            elif dist == "tiny":
                lens = rng.integers(1, min(t, 4) + 1, size=b)
            elif dist == "one_long":
                lens = rng.integers(1, max(2, t // 8) + 1, size=b); lens[int(rng.integers(b))] = t
            elif dist == "mult32":
                lens = np.minimum(t, 32 * rng.integers(1, max(2, t // 32 + 1), size=b))
            elif dist == "off32":
                lens = np.clip(32 * rng.integers(1, max(2, t // 32 + 1), size=b) + rng.integers(-1, 2, size=b), 1, t)
            else:
                lens = rng.integers(1, t + 1, size=b)
            lens = np.minimum(lens, t).astype(np.int64)
            ids = rng.integers(5, 900, size=(b, t)).astype(np.int64)
            mask = (np.arange(t)[None, :] < lens[:, None]).astype(np.int64)
            ids[mask == 0] = pad_id
            with torch.no_grad():
                if kind.startswith("bert"):
                    ref = m(input_ids=torch.from_numpy(ids).to(dev), attention_mask=torch.from_numpy(mask).to(dev)).last_hidden_state[:, 0]
                    ref = torch.nn.functional.normalize(ref, dim=-1).cpu().numpy()
                else:
                    ref = m(input_ids=torch.from_numpy(ids).to(dev), attention_mask=torch.from_numpy(mask).to(dev)).logits[:, 0].cpu().numpy()
            got = encs[kind].forward(ids.astype(np.int32), mask.astype(np.int32))
            again = encs[kind].forward(ids.astype(np.int32), mask.astype(np.int32))      # the same call twice: bit for bit the same
            if not np.array_equal(np.asarray(got).view(np.uint32), np.asarray(again).view(np.uint32)):
                fails += 1
                print("FAIL (not repeatable)", json.dumps({"kind": kind, "b": b, "t": t, "dist": dist, "lens": lens.tolist()[:50]}), flush=True)
            err = float(np.abs(np.asarray(got).reshape(ref.shape) - ref).max()) if np.isfinite(got).all() else float("inf")
            if not kind.startswith("bert"):
                err /= max(1.0, float(np.abs(ref).max()))          # logits: relative to their scale, as the tests do
            worst[kind] = max(worst.get(kind, 0.0), err / tol)
            n += 1
            if not err < tol:
                fails += 1
                print("FAIL", json.dumps({"kind": kind, "b": b, "t": t, "dist": dist, "lens": lens.tolist()[:50], "err": err, "tol": tol}), flush=True)
            if n % 50 == 0:
                print(f"... {n} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    finally:
        for e in encs.values():
            e.close()
    print(json.dumps({"cases": n, "failures": fails, "worst_err_over_tol": {k: round(v, 3) for k, v in worst.items()},
                      "seconds": round(time.time() - t0, 1), "seed": a.seed}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
