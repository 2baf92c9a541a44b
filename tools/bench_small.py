#!/usr/bin/env python3
"""Latency of the reference's real call shape (ensembleRetriever.py:64-66): N ~ 1e4 chunks, d = 1024 (bge-m3),
nq = 1..4 query strings (query + HyDE), k = 2048, host buffers in / out (FaissRetriever.invoke after embedding)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import veritasfi_amd as vf
rng = np.random.default_rng(0)
for n in (2_000, 10_000, 16_000, 50_000):
    c = rng.standard_normal((n, 1024)).astype(np.float32)
    ix = vf.DenseIndex(c)
    for nq in (1, 4):
        q = rng.standard_normal((nq, 1024)).astype(np.float32)
        ix.search(q, 2048)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); ix.search(q, 2048); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"n={n:6d} nq={nq} k=2048 path={ix.stats()['path']}: p50 {np.median(ts):.3f} ms  min {min(ts):.3f} ms", flush=True)
    ix.close()
