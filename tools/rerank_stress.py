#!/usr/bin/env python3
"""Counterpart of the reference's re-rank stress worker (experiments/profile/stress_test.py: model_worker :187-244,
statistics :164-185): loop { retrieve the top chunks for a random question -> score (question, chunk) pairs in
batches of 8 -> chunk similarity matrix }, reporting inference calls / second after a warm-up.  Synthetic corpus,
random-init encoder + cross-encoder of the named shapes; every stage through this package's drop-ins
(FaissRetriever.invoke, HipReranker.compute_score, compute_similarity_mtx)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import veritasfi_amd as vf
from _synth import HashTokenizer, sentence
from bench_rerank import random_encoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--embedder", default="bert-base")
    ap.add_argument("--reranker", default="xlmr-base")
    ap.add_argument("--corpus", type=int, default=5000)
    ap.add_argument("--pairs", type=int, default=155, help="chunks retrieved and scored per call (stress_test.py:152)")
    ap.add_argument("--words", type=int, default=200)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--warmup_s", type=float, default=5.0)
    ap.add_argument("--seconds", type=float, default=20.0)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    e_enc, e_cfg = random_encoder(a.embedder, head=0)
    r_enc, r_cfg = random_encoder(a.reranker, head=1)
    emb = vf.HipEmbeddings(HashTokenizer(e_cfg["vocab"]), e_enc, max_length=512, batch_size=100)
    rr = vf.HipReranker(HashTokenizer(r_cfg["vocab"]), r_enc, max_length=512)
    docs = [sentence(rng, a.words) for _ in range(a.corpus)]
    questions = [sentence(rng, 14) for _ in range(32)]
    t0 = time.time()
    vecs = emb.embed_documents(docs)                       # the embed loop (load_data.py:120-128), batches of 100
    print(f"embedded {len(docs)} chunks in {time.time() - t0:.2f}s ({len(docs) / (time.time() - t0):.0f} chunks/s)", flush=True)
    fr = vf.FaissRetriever(vecs, emb)
    calls, t_start, t_end, lat = 0, None, time.time() + a.warmup_s + a.seconds, []
    warm_until = time.time() + a.warmup_s
    while time.time() < t_end:
        c0 = time.time()
        q = questions[int(rng.integers(len(questions)))]
        I, _ = fr.invoke([q], a.pairs)
        chunks = [docs[i] for i in I[0] if i >= 0]
        scores = rr.compute_score([[q, c] for c in chunks], batch_size=a.batch_size)
        mtx = vf.compute_similarity_mtx(emb, chunks)
        assert len(scores) == len(chunks) and tuple(mtx.shape) == (len(chunks), len(chunks))
        if time.time() >= warm_until:
            if t_start is None:
                t_start = c0
            calls += 1
            lat.append(time.time() - c0)
    el = time.time() - (t_start or time.time())
    print(f"Total inference calls: {calls}")
    print(f"Elapsed time: {el:.2f} seconds")
    print(f"Overall rate: {calls / max(el, 1e-9):.2f} inference calls/second")
    if lat:
        print(f"Latency per call: p50 {np.median(lat) * 1e3:.1f} ms  p90 {np.percentile(lat, 90) * 1e3:.1f} ms "
              f"({a.pairs} pairs scored in batches of {a.batch_size} + {a.pairs}x{a.pairs} similarity matrix)")
    e_enc.close(); r_enc.close()


if __name__ == "__main__":
    main()
