"""Control-flow rows against the REAL reference's outputs (tests/golden/g5-g7, written by tools/gen_golden.py from
src/utils/ensembleRetriever.py, src/utils/vllmManager.py and experiments/profile/stress_test.py running in the build
container).  Two things are pinned to each fixture: the oracle's literal restatement (oracle/ref_ensemble.py,
oracle/ref_numpy.rank_chunk, oracle/ref_rerank_inputs.py) and the product (veritasfi_amd.EnsembleRetriever, rank_chunk,
build_llm_reranker_inputs).  The GPU cases run the product through the HIP library."""
import json
import os
import warnings
from datetime import datetime

import numpy as np
import pytest

import golden_inputs as GI
from conftest import GOLDEN, load_golden


def _g5():
    with open(os.path.join(GOLDEN, "g5_ensemble_invoke.json")) as f:
        return json.load(f)["cases"]


def _g6():
    with open(os.path.join(GOLDEN, "g6_rank_chunk.json")) as f:
        return json.load(f)["cases"]


def _g5_world(rec):
    seed, n, k, faiss_k, faiss_ts_k, bm25_k, expand = rec["case"]
    w = GI.g5_world(seed, n=n)
    sha = GI.sha(w["embs"], w["t_embs"], w["bm_scores"], np.array(w["bm_order"]),
                 np.frombuffer(json.dumps([w["metas"], w["titles"], w["queries"]], sort_keys=True).encode(), np.uint8))
    assert sha == rec["input_sha"], "golden inputs drifted"
    return w


def _expected(rec, w, qi):
    return [{"retriever": name, "score": score, "page_content": w["docs"][row], "metadata": w["metas"][row],
             "bundle_id": b} for name, score, row, b in rec["outputs"][qi]]


def _same_chunks(got, want, score_tol):
    assert [(c["retriever"], c["page_content"], c["metadata"], c["bundle_id"]) for c in got] == \
           [(c["retriever"], c["page_content"], c["metadata"], c["bundle_id"]) for c in want]
    assert all(type(c["score"]) is float and set(c) == {"retriever", "score", "page_content", "metadata", "bundle_id"}
               for c in got)
    if score_tol == 0:
        assert [c["score"] for c in got] == [c["score"] for c in want]
    elif got:
        assert max(abs(a["score"] - b["score"]) for a, b in zip(got, want)) <= score_tol


@pytest.mark.parametrize("ci", range(len(GI.G5_CASES)))
def test_g5_restatement_matches_reference_invoke(ci):
    """oracle/ref_ensemble.gather == the real EnsembleRetriever.invoke (ensembleRetriever.py:50-233), exactly."""
    from oracle import ref_ensemble as RE
    rec = _g5()[ci]
    w = _g5_world(rec)
    seed, n, k, faiss_k, faiss_ts_k, bm25_k, expand = rec["case"]
    emb = GI.TableEmbeddings(w["table"])
    store = GI.Store(w["docs"], w["metas"], None)
    dense, ts_dense = GI.CosineRetriever(w["embs"], emb), GI.CosineRetriever(w["t_embs"], emb)
    bm = GI.ListBM25(w["bm_order"], w["bm_scores"])
    for qi, (q, hyde) in enumerate(w["queries"]):
        got = RE.gather(q, hyde, chunk_metadata=w["metas"], title_summaries=w["titles"],
                        store_get=lambda ids: store.get(ids=ids, include=["documents", "metadatas"]),
                        dense=dense.invoke, ts_dense=ts_dense.invoke, bm25=bm.invoke,
                        faiss_k=k if faiss_k is None else faiss_k, faiss_ts_k=faiss_ts_k, bm25_k=bm25_k, enable_expand=expand)
        _same_chunks(got, _expected(rec, w, qi), 0)


def _product_retriever(rec, w, **kw):
    from veritasfi_amd.ensemble import EnsembleRetriever
    seed, n, k, faiss_k, faiss_ts_k, bm25_k, expand = rec["case"]
    chroma = GI.Store(w["docs"], w["metas"], w["embs"].tolist())
    ts = GI.Store(w["titles"], [None] * len(w["titles"]), w["t_embs"].tolist())
    return EnsembleRetriever("bm25_dir", chroma, ts, k, GI.TableEmbeddings(w["table"]), faiss_k=faiss_k, bm25_k=bm25_k,
                             faiss_ts_k=faiss_ts_k, enable_expand=expand,
                             bm25_retriever=GI.ListBM25(w["bm_order"], w["bm_scores"]), **kw)


@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("ci", range(len(GI.G5_CASES)))
def test_g5_product_host_logic_matches_reference_invoke(ci, prefetch):
    """The pre-indexed product class on the SAME injected exact-cosine retriever the fixture was generated with:
    identical output, scores included."""
    rec = _g5()[ci]
    w = _g5_world(rec)
    er = _product_retriever(rec, w, retriever_cls=GI.CosineRetriever, prefetch_documents=prefetch)
    for qi, (q, hyde) in enumerate(w["queries"]):
        _same_chunks(er.invoke(q, hyde), _expected(rec, w, qi), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(GI.G5_CASES)))
def test_g5_gpu_ensemble_matches_reference_invoke(ci):
    """EnsembleRetriever over the HIP FaissRetriever (vf_index_create / vf_index_search, k = 2048) against the real
    reference's invoke: same retriever labels, chunks, metadata, bundle ids and order; scores within 1e-5 (the fixture's
    dense scores are NumPy sgemm, the product's are canonical; every compared score is >= `threshold_margin` away from
    0.72 / 0.66, recorded at generation)."""
    rec = _g5()[ci]
    assert rec["threshold_margin"] > 2e-5
    w = _g5_world(rec)
    er = _product_retriever(rec, w)
    from veritasfi_amd.faiss_retriever import FaissRetriever
    assert isinstance(er.faiss_retriever, FaissRetriever)
    for qi, (q, hyde) in enumerate(w["queries"]):
        _same_chunks(er.invoke(q, hyde), _expected(rec, w, qi), 1e-5)


def _g6_args(ci, rec):
    inp = GI.g6_inputs(ci)
    assert GI.sha(np.frombuffer(json.dumps(inp, sort_keys=True).encode(), np.uint8)) == rec["input_sha"]
    return inp


@pytest.mark.parametrize("ci", range(len(GI.G6_CASES)))
def test_g6_restatement_matches_reference_rank_chunk(ci):
    """oracle/ref_numpy.rank_chunk == the real ChatManager.rank_chunk (vllmManager.py:430-483)."""
    from oracle import ref_numpy as R
    rec = _g6()[ci]
    inp = _g6_args(ci, rec)
    chunks = inp["chunks"]
    qt = datetime(*inp["query_time"])
    days = [(qt - datetime.strptime(c["metadata"]["date_published"], "%Y-%m-%d")).days for c in chunks]
    args = ([c["bundle_id"] for c in chunks], [inp["rr"][c["page_content"]] for c in chunks], R.time_scores(np.array(days)),
            np.array([inp["emb"][c["page_content"]] for c in chunks], np.float32), inp["chunk_topk"])
    if "raises" in rec:
        with pytest.raises(IndexError):
            R.rank_chunk(*args)
        assert rec["raises"] == "IndexError"
    else:
        assert R.rank_chunk(*args) == rec["selected"]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(GI.G6_CASES)))
def test_g6_gpu_rank_chunk_matches_reference(ci):
    """veritasfi_amd.rank_chunk (vf_fuse_rank + vf_cosine_matrix on the GPU, the greedy loop on the host) returns the
    real reference's selection, including the bundle-id-as-column quirk of :476 and its out-of-range failure."""
    import veritasfi_amd as vf
    rec = _g6()[ci]
    inp = _g6_args(ci, rec)

    class Reranker:
        def compute_score(self, pairs, batch_size=8):
            return [inp["rr"][p[1]] for p in pairs]

    call = lambda: vf.rank_chunk(inp["chunks"], inp["question"], datetime(*inp["query_time"]), Reranker(),
                                 GI.TableEmbeddings(inp["emb"]), inp["chunk_topk"])
    assert rec["sim_margin"] > 1e-4
    if "raises" in rec:
        with pytest.raises(IndexError):
            call()
    else:
        assert [int(b) for b in call()] == rec["selected"]


@pytest.mark.parametrize("ci", range(len(GI.G7_CASES)))
def test_g7_rerank_inputs_match_reference_get_inputs(ci):
    """oracle/ref_rerank_inputs.get_inputs and the product's build_llm_reranker_inputs (+ HipLLMReranker's own padding
    rule) against the real get_inputs (stress_test.py:97-146) on transformers' PreTrainedTokenizer."""
    from oracle import ref_rerank_inputs as RI
    from veritasfi_amd.encoder import build_llm_reranker_inputs
    g = load_golden("g7_rerank_get_inputs.npz")
    pairs, max_length, side = GI.g7_pairs(ci)
    assert str(g[f"sha{ci}"]) == GI.sha(np.frombuffer(json.dumps([pairs, max_length, side]).encode(), np.uint8))
    ids, mask = g[f"ids{ci}"], g[f"mask{ci}"]
    tok = GI.g7_tokenizer(side)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # transformers: "max_length is ignored when padding=True" (the reference's call)
        ref = RI.get_inputs(pairs, tok, max_length=max_length)
    assert np.array_equal(ref["input_ids"], ids) and np.array_equal(ref["attention_mask"], mask)
    rows = build_llm_reranker_inputs(pairs, tok, max_length=max_length)
    assert len(rows) == ids.shape[0]
    for i, row in enumerate(rows):
        assert row == ids[i][mask[i] == 1].tolist()
    assert -(-max(len(r) for r in rows) // 8) * 8 == ids.shape[1]          # pad_to_multiple_of=8
    assert max(len(r) for r in rows) > max_length or max_length >= 1024     # clipped pair + separator + prompt

