#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE in the build container.

Run from the repo root:  python tools/gen_golden.py
Needs /root/reference (read-only) -- it never travels to the GPU box; only the small .npz
fixtures written here do.  A fixture holds inputs and the reference's outputs, nothing else.

What is imported:
  * experiments/retriever/continuous_retrieval.py  -- imports as-is.
  * experiments/retriever/step3_mul.py             -- needs a stand-in for the absent `openai`
    package in sys.modules (the LLM judge client; not on the hot path, never called here).
The encoder the reference would load (Qwen3-Embedding via transformers) has no weights here, so a
fake tokenizer/model pair maps the text "i" to a fixed embedding row i; everything downstream of
get_embeddings -- sklearn.cosine_similarity, np.argsort, slicing, output types -- is the reference's
own code running unmodified.

Inputs come from tests/golden_inputs.py (seeded); a fixture stores the sha256 of the input bytes
plus the REFERENCE's outputs, so the fixtures stay small and a generator drift fails loudly.
Adjacent reference scores get as close as 3e-8 at k=100 (recorded as `min_gap`): ids inside such
near-tie groups depend on the BLAS summation order, so tests compare per near-tie group.
  G1  continuous_retrieval.select_top_chunks       N=1000, d=768, k in {3, 8}
  G2  step3_mul.select_top_chunks_batch            (E, C, d, k) grid incl. k=-1 (full sort)
  G3  raw sklearn cosine_similarity on fp16-rounded inputs (C2-like distribution)
  G4  tie characterisation: duplicate / scaled rows -> reference's order + the tie groups
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/experiments/retriever"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402


class FakeTokenizer:
    """text "i" -> input_ids [[i]], attention_mask [[1]] (so both pooling variants pick that row)."""

    def __call__(self, texts, padding=True, truncation=True, return_tensors="pt", max_length=None):
        ids = torch.tensor([[int(t)] for t in texts], dtype=torch.long)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}


class FakeModel:
    def __init__(self, table: np.ndarray):
        self.table = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32))

    def __call__(self, input_ids=None, attention_mask=None, **kw):
        return types.SimpleNamespace(last_hidden_state=self.table[input_ids])  # [b, 1, d]


def import_reference():
    sys.path.insert(0, REF)
    saved = os.environ.get("CUDA_VISIBLE_DEVICES")
    import continuous_retrieval as cr  # noqa: E402  (sets CUDA_VISIBLE_DEVICES as a side effect)
    if "openai" not in sys.modules:
        stub = types.ModuleType("openai")
        stub.OpenAI = object
        sys.modules["openai"] = stub
    import step3_mul as s3  # noqa: E402
    if saved is None:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    else:
        os.environ["CUDA_VISIBLE_DEVICES"] = saved
    return cr, s3


def min_gap(vals):
    v = np.asarray(vals, dtype=np.float64)
    return float(np.min(np.abs(np.diff(v)))) if v.size > 1 else 1.0


def main():
    os.makedirs(OUT, exist_ok=True)
    cr, s3 = import_reference()
    tok = FakeTokenizer()
    dev = torch.device("cpu")

    # ---- G1: continuous_retrieval.select_top_chunks (returns chunk strings only)
    chunks, evid = GI.g1_inputs()
    n = chunks.shape[0]
    model = FakeModel(np.vstack([chunks, evid]))
    texts = [str(i) for i in range(n)]
    g1 = {"input_sha": np.array(GI.sha(chunks, evid))}
    for k in (3, 8):
        top = cr.select_top_chunks(str(n), texts, model, tok, dev, top_k=k, batch_size=32)
        g1[f"ids_k{k}"] = np.array([int(t) for t in top], dtype=np.int64)
    # the reference's own embedding round trip (mean-pool of a length-1 sequence == the row)
    emb = cr.get_embeddings(texts[:5], model, tok, dev, batch_size=2)
    assert emb.dtype == np.float32 and np.array_equal(emb, chunks[:5])
    g1["sim_row"] = cr.cosine_similarity(evid, chunks)[0]
    np.savez_compressed(os.path.join(OUT, "g1_continuous_select_top_chunks.npz"), **g1)

    # ---- G2: step3_mul.select_top_chunks_batch
    for ci, (E, C, d, k) in enumerate(GI.G2_CASES):
        chunks, evid, _ = GI.g2_inputs(ci)
        model = FakeModel(np.vstack([chunks, evid]))
        res = s3.select_top_chunks_batch([str(C + i) for i in range(E)], [str(i) for i in range(C)],
                                         model, tok, dev, top_k=k, batch_size=16)
        kk = C if k == -1 else k
        ids = np.array([[int(t) for t in r[0]] for r in res], dtype=np.int64)
        sims = np.array([[np.float32(v) for v in r[1]] for r in res], dtype=np.float32)
        assert ids.shape == (E, kk) and all(isinstance(v, np.float32) for v in res[0][1])
        gaps = np.array([min_gap(s) for s in sims])
        # single-evidence variant must agree with the batch variant on scores (same file :233-253;
        # ids may differ inside near-ties because sgemv and sgemm sum in different orders)
        one_chunks, one_sims = s3.select_top_chunks(str(C), [str(i) for i in range(C)], model, tok, dev,
                                                    top_k=k, batch_size=16)
        assert np.allclose(np.array(one_sims, dtype=np.float32), sims[0], atol=1e-6)
        np.savez_compressed(os.path.join(OUT, f"g2_step3_batch_case{ci}.npz"),
                            input_sha=np.array(GI.sha(chunks, evid)), k=np.int64(k), ids=ids, sims=sims,
                            min_gap=gaps)
        print(f"G2 case{ci} E={E} C={C} d={d} k={k}: min adjacent gap {gaps.min():.3e}")

    # ---- G3: raw cosine_similarity on fp16-rounded C2-like inputs
    corpus, queries = GI.g3_inputs()
    sim = s3.cosine_similarity(queries, corpus.astype(np.float32))
    assert sim.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "g3_cosine_fp16_inputs.npz"), input_sha=np.array(GI.sha(corpus, queries)),
                        sim=sim)

    # ---- G4: ties (characterisation only; reference order among exact ties is implementation-defined)
    chunks, evid, groups = GI.g4_inputs()
    model = FakeModel(np.vstack([chunks, evid]))
    C = chunks.shape[0]
    res = s3.select_top_chunks_batch([str(C)], [str(i) for i in range(C)], model, tok, dev, top_k=-1, batch_size=16)
    np.savez_compressed(os.path.join(OUT, "g4_ties.npz"), input_sha=np.array(GI.sha(chunks, evid)),
                        ids=np.array([int(t) for t in res[0][0]], dtype=np.int64),
                        sims=np.array(res[0][1], dtype=np.float32))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
