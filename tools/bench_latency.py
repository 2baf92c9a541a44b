#!/usr/bin/env python3
"""Latency of the reference's per-request call shapes: one embed_query forward (faissRetriever.py:33: one query string,
<= 32 / 64 tokens) and FaissRetriever.invoke's search (ensembleRetriever.py:64-66: N ~ 1e4, d = 1024, nq <= 4, k = 2048),
host buffers in and out.  Prints one JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import veritasfi_amd as vf
from bench_rerank import random_encoder


def p50(fn, n=40):
    fn(); fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 4), round(float(min(ts)), 4)


def main():
    rng = np.random.default_rng(0)
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    for shape in ("bert-base", "xlmr-large"):
        enc, cfg = random_encoder(shape, head=0)
        for t in (32, 64):
            ids = rng.integers(5, cfg["vocab"], size=(1, t)).astype(np.int32)
            mask = np.ones_like(ids)
            m, lo = p50(lambda: enc.forward(ids, mask))
            print(json.dumps({"case": "embed_query", "model_shape": shape, "tokens": t, "p50_ms": m, "min_ms": lo}), flush=True)
        enc.close()
    c = rng.standard_normal((10_000, 1024)).astype(np.float32)
    ix = vf.DenseIndex(c)
    for nq in (1, 4):
        q = rng.standard_normal((nq, 1024)).astype(np.float32)
        m, lo = p50(lambda: ix.search(q, 2048))
        print(json.dumps({"case": "invoke_search", "n": 10_000, "d": 1024, "nq": nq, "k": 2048, "p50_ms": m, "min_ms": lo}), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
