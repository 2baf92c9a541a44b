#!/usr/bin/env python3
"""Re-rank forward on RAGGED batches (pair lengths uniform in [lo, hi], right-padded to the longest): the packed forward
(default) vs the padded one (VF_NO_PACKED=1, decided per process).  Random-weight XLM-R-base shape cross-encoder."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from bench_rerank import random_encoder

def main():
    enc, cfg = random_encoder("xlmr-base", head=1, vocab=4096)
    rng = np.random.default_rng(1)
    for (n, lo, hi) in ((100, 128, 512), (100, 64, 512), (100, 300, 512), (100, 512, 512), (32, 40, 256)):
        lens = rng.integers(lo, hi + 1, size=n)
        t = int(-(-lens.max() // 32) * 32)
        ids = rng.integers(5, cfg["vocab"], size=(n, t)).astype(np.int32)
        mask = (np.arange(t)[None, :] < lens[:, None]).astype(np.int32)
        ids[mask == 0] = 1
        enc.forward(ids, mask); enc.forward(ids, mask)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); out = enc.forward(ids, mask); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"packed": os.environ.get("VF_NO_PACKED") is None, "pairs": n, "tokens": f"{lo}..{hi}",
                          "valid_tokens": int(lens.sum()), "padded_tokens": n * t, "p50_ms": round(float(np.median(ts)), 3),
                          "score_checksum": float(np.abs(out).sum())}), flush=True)
    enc.close()

if __name__ == "__main__":
    main()
