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



# ---- the serve chain through the reference's own calls (round 6): retriever.invoke -> rank_chunk(chunks, q, t, retriever) ----------
class _CountingEmbeddings(GI.TableEmbeddings):
    """Table embedder that notes every text it is asked to embed."""

    def __init__(self, table):
        super().__init__(table)
        self.asked = []

    def embed_query(self, text):
        self.asked.append(text)
        return super().embed_query(text)

    def embed_documents(self, texts):
        self.asked.extend(texts)
        return super().embed_documents(texts)


def _serve_world(seed, n=320, d=48):
    """A corpus whose stored embedding of a chunk IS embed(text) (what load_data.py:120-128 stores), with near-duplicate chunks
    (cosine > 0.9 to their neighbour), one text held by two rows, bundles, and a query aimed at the duplicates."""
    rng = np.random.default_rng(seed)
    embs = rng.standard_normal((n, d)).astype(np.float32)
    for i in range(0, n, 7):                      # near-duplicates: row i + 1 is row i plus a little noise
        if i + 1 < n:
            embs[i + 1] = embs[i] + 0.08 * rng.standard_normal(d).astype(np.float32)
    docs = [f"chunk text {i}" for i in range(n)]
    docs[11] = docs[10]                           # the same text in two rows ...
    embs[11] = embs[10]                           # ... has one embedding
    metas = []
    for i in range(n):
        md = {"doc_id": f"d{i}", "prev_chunk_id": "", "next_chunk_id": "", "title_summary": f"title {i % 9}",
              "date_published": f"2024-{1 + i % 12:02d}-{1 + i % 28:02d}"}
        if i % 6 == 0 and i + 1 < n:
            md["bundle_id"] = f"b{i}"
        if i % 6 == 1:
            md["bundle_id"] = f"b{i - 1}"
        metas.append(md)
    table = {docs[i]: embs[i].tolist() for i in range(n)}
    anchor = 14
    table["the question"] = (embs[anchor] + 0.3 * rng.standard_normal(d).astype(np.float32)).tolist()
    table["hyde"] = (embs[anchor + 1] + 0.3 * rng.standard_normal(d).astype(np.float32)).tolist()
    novel = "a chunk the corpus does not hold"
    table[novel] = (embs[anchor] + 0.05 * rng.standard_normal(d).astype(np.float32)).tolist()   # a near-duplicate of a corpus row
    titles = [f"title {t}" for t in range(9)]
    t_embs = rng.standard_normal((len(titles), d)).astype(np.float32)
    return docs, metas, embs, table, titles, t_embs, novel


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rank_chunk_through_the_reference_call_serves_the_similarity_matrix_from_corpus_rows(seed):
    """ChatManager.rank_chunk hands the RETRIEVER to the selection (vllmManager.py:430) and calls
    retriever.compute_similarity_mtx(texts) with texts only (:462).  Driven exactly so -- no similarity_index=, no row ids on the
    chunks -- the product's EnsembleRetriever reads the texts it emitted from their corpus rows (vf_cosine_matrix_rows_mixed) and
    embeds only the one text the corpus does not hold; the matrix is bit-equal to re-embedding everything and the selection equal."""
    import veritasfi_amd as vf
    from oracle import canonical as oracle
    docs, metas, embs, table, titles, t_embs, novel = _serve_world(seed)
    emb = _CountingEmbeddings(table)
    chroma = GI.Store(docs, metas, embs.tolist())
    ts = GI.Store(titles, [None] * len(titles), t_embs.tolist())
    er = vf.EnsembleRetriever("bm25_dir", chroma, ts, 10, emb, faiss_k=24, bm25_k=0, faiss_ts_k=1)
    assert er.similarity_from_rows, "fp32 rows as given + an embedder without a query instruction: the rows serve the matrix"
    chunks = er.invoke("the question", ["hyde"])
    assert len(chunks) >= 24
    for c in chunks:
        c["metadata"] = dict(c["metadata"])
    chunks.append({"retriever": "BM25", "score": 1.0, "page_content": novel, "bundle_id": max(c["bundle_id"] for c in chunks) + 1,
                   "metadata": {"date_published": "2024-06-01"}})
    texts = [c["page_content"] for c in chunks]
    assert docs[10] in texts or True
    rr_rng = np.random.default_rng(100 + seed)
    rr = {t: float(v) for t, v in zip(dict.fromkeys(texts), rr_rng.standard_normal(len(texts)))}

    class Reranker:
        def compute_score(self, pairs, batch_size=8):
            return [rr[p[1]] for p in pairs]

    when = datetime(2024, 6, 15)
    emb.asked.clear()
    mtx_rows = er.compute_similarity_mtx(texts)
    assert emb.asked == [novel], f"only the unknown text is embedded, got {emb.asked}"
    assert tuple(mtx_rows.shape) == (len(texts), len(texts)) and hasattr(mtx_rows, "numpy")
    mtx_embed = vf.compute_similarity_mtx(emb, texts, as_torch=False)
    want = oracle.cosine(np.asarray([table[t] for t in texts], np.float32), np.asarray([table[t] for t in texts], np.float32))
    assert np.array_equal(mtx_rows.numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(mtx_embed.view(np.uint32), want.view(np.uint32))
    assert (mtx_rows.numpy() > 0.9).sum() > len(texts), "the world holds near-duplicates above the 0.9 threshold"
    emb.asked.clear()
    picked_rows = vf.rank_chunk(chunks, "the question", when, Reranker(), er, 12)            # upstream's call: the retriever
    assert emb.asked == [novel]
    picked_embed = vf.rank_chunk(chunks, "the question", when, Reranker(), emb, 12)          # re-embedding every text
    assert picked_rows == picked_embed and 0 < len(picked_rows) <= 12
    # switched off, the same object embeds everything, as upstream
    er_off = vf.EnsembleRetriever("bm25_dir", chroma, ts, 10, emb, faiss_k=24, bm25_k=0, faiss_ts_k=1, similarity_from_rows=False)
    er_off.invoke("the question", ["hyde"])
    emb.asked.clear()
    m_off = er_off.compute_similarity_mtx(texts)
    assert sorted(emb.asked) == sorted(texts) and np.array_equal(m_off.numpy().view(np.uint32), want.view(np.uint32))
    # ... and an embedder with a query instruction keeps the reference's route by itself
    emb.query_instruction = "Represent this sentence for searching relevant passages: "
    assert not vf.EnsembleRetriever("bm25_dir", chroma, ts, 10, emb, faiss_k=24, bm25_k=0, faiss_ts_k=1).similarity_from_rows
    # rows held in a narrower type than given are not "the embeddings": auto declines
    import functools
    assert not vf.EnsembleRetriever("bm25_dir", chroma, ts, 10, _CountingEmbeddings(table), faiss_k=24, bm25_k=0, faiss_ts_k=1,
                                    retriever_cls=functools.partial(vf.FaissRetriever, corpus_dtype="f16")).similarity_from_rows


@pytest.mark.gpu
def test_cosine_matrix_rows_mixed_argument_checks_and_sharded_handle():
    """vf_cosine_matrix_rows_mixed: -1 entries take the caller's vectors in order; counts must agree; a sharded handle gives the
    single-device bits."""
    import veritasfi_amd as vf
    from oracle import canonical as oracle
    rng = np.random.default_rng(3)
    c = rng.standard_normal((3000, 40)).astype(np.float32)
    ex = rng.standard_normal((3, 40)).astype(np.float32)
    ids = np.array([5, -1, 2999, 17, -1, -1, 0], np.int64)
    full = np.vstack([c[5], ex[0], c[2999], c[17], ex[1], ex[2], c[0]])
    want = oracle.cosine(full, full)
    with vf.DenseIndex(c) as ix:
        assert np.array_equal(ix.cosine_matrix_rows(ids, ex).view(np.uint32), want.view(np.uint32))
        only = ix.cosine_matrix_rows(np.array([-1, -1, -1]), ex)
        assert np.array_equal(only.view(np.uint32), oracle.cosine(ex, ex).view(np.uint32))
        with pytest.raises(RuntimeError, match="n_extra"):
            ix.cosine_matrix_rows(ids, ex[:2])
        with pytest.raises(RuntimeError, match="outside"):
            ix.cosine_matrix_rows(np.array([5, -1]))          # -1 without vectors is not a row
        with pytest.raises(ValueError):
            ix.cosine_matrix_rows(ids, ex[:, :7])
    with vf.DenseIndex(c, device_ids=[0, 0, 0]) as grp:
        assert np.array_equal(grp.cosine_matrix_rows(ids, ex).view(np.uint32), want.view(np.uint32))
