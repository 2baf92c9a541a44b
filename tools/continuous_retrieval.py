#!/usr/bin/env python3
"""Counterpart of the reference's retrieval throughput harness (experiments/retriever/continuous_retrieval.py:
process_sample :69-118, print_statistics :169-190, iteration loop :251-273) on synthetic samples: per sample,
for each evidence string, embed evidence + chunks (mean-pool get_embeddings) and keep the top-3 chunks by cosine.
Every stage runs on the GPU through this package's drop-ins; the reference's worker pool over GPUs maps to
replicas (one such process per GPU), so a single process is timed here."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import veritasfi_amd as vf
from _synth import HashTokenizer, sentence
from bench_rerank import random_encoder


def make_dataset(n, evidences, chunks, words, seed=0):
    rng = np.random.default_rng(seed)
    return [{"question": sentence(rng, 12), "evidence": [sentence(rng, words) for _ in range(evidences)],
             "query_chunks": [sentence(rng, words) for _ in range(chunks)]} for _ in range(n)]


def process_sample(example, model, tokenizer, batch_size):
    start = time.time()
    evidence_list, query_chunks = example.get("evidence", []), example.get("query_chunks", [])
    if not evidence_list or not query_chunks:
        return {"status": "skipped", "num_evidences": 0, "num_chunks": 0, "retrieval_time": 0.0}
    total = 0
    for evidence in evidence_list:
        top = vf.select_top_chunks(evidence, query_chunks, model, tokenizer, "cpu", top_k=3, batch_size=batch_size,
                                   pooling="mean", return_similarities=False)
        total += len(top)
    return {"status": "success", "num_evidences": len(evidence_list), "num_chunks": len(query_chunks),
            "total_retrievals": total, "retrieval_time": time.time() - start}


def print_statistics(results, elapsed):
    ok = [r for r in results if r["status"] == "success"]
    if not ok:
        print("No successful retrievals")
        return
    n = len(ok)
    print("=" * 60)
    print("Retrieval Statistics")
    print("=" * 60)
    print(f"Total samples processed: {n}")
    print(f"Total retrievals: {sum(r['total_retrievals'] for r in ok)}")
    print(f"Average retrieval time per sample: {sum(r['retrieval_time'] for r in ok) / n:.4f}s")
    print(f"Throughput: {n / elapsed:.2f} samples/sec")
    print(f"Total elapsed time: {elapsed:.2f}s")
    print("=" * 60, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="bert-base")
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--evidences", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=60)
    ap.add_argument("--words", type=int, default=120, help="words per text (~tokens)")
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--iterations", type=int, default=2)
    a = ap.parse_args()
    enc, cfg = random_encoder(a.shape, head=0)
    model, tok = vf.HipModel(enc), HashTokenizer(cfg["vocab"])
    data = make_dataset(a.samples, a.evidences, a.chunks, a.words)
    for it in range(1, a.iterations + 1):
        print(f"Starting iteration {it}", flush=True)
        t0 = time.time()
        results = [process_sample(ex, model, tok, a.batch_size) for ex in data]
        print_statistics(results, time.time() - t0)
    enc.close()


if __name__ == "__main__":
    main()
