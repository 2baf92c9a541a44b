#!/usr/bin/env python3
"""Experiment: do two half-batches of a re-rank forward, run CONCURRENTLY on two streams, finish sooner than the whole batch
on one?  (The 256x256 GEMM tiles of 100 x 512 rows leave partial last rounds: 7.03 / 2.34 / 9.4 / 2.34 rounds of 256 CUs.)
Needs a build with -fgpu-default-stream=per-thread (VF_LIB_PATH): two handles, two Python threads."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from bench_rerank import random_encoder

pairs, tokens, shape = 100, 512, sys.argv[1] if len(sys.argv) > 1 else "xlmr-base"
encs = [random_encoder(shape, 1, vocab=2000)[0] for _ in range(2)]
rng = np.random.default_rng(1)
ids = rng.integers(5, 2000, size=(pairs, tokens)).astype(np.int32)
mask = np.ones_like(ids)
half = pairs // 2

def whole():
    encs[0].forward(ids, mask)

def halves():
    bar = threading.Barrier(2)
    def run(i):
        bar.wait()
        encs[i].forward(ids[i * half:(i + 1) * half], mask[i * half:(i + 1) * half])
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()

def p50(fn, n=15):
    fn(); fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)

print(json.dumps({"shape": shape, "whole_batch_ms": p50(whole), "two_concurrent_halves_ms": p50(halves),
                  "one_half_alone_ms": p50(lambda: encs[0].forward(ids[:half], mask[:half]))}))
