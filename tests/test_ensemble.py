"""EnsembleRetriever (pre-indexed bundle / title / doc-id maps) against the literal restatement of
src/utils/ensembleRetriever.py:50-232, on the same retriever outputs.  CPU-only: the dense retrievers are injected
(the GPU FaissRetriever itself is covered by tests/test_gpu_retrieval.py; the GPU end-to-end case is below)."""
import numpy as np
import pytest

from oracle import ref_ensemble as RE


class Store:
    """Chroma-shaped store: get(include=[...]) returns everything, get(ids=[...]) the requested rows in order."""

    def __init__(self, docs, metas, embs):
        self.docs, self.metas, self.embs = docs, metas, embs
        self.by_id = {m["doc_id"]: i for i, m in enumerate(metas)} if metas and metas[0] else {}
        self.calls = 0

    def get(self, ids=None, include=()):
        if ids is None:
            return {"documents": self.docs, "metadatas": self.metas, "embeddings": self.embs}
        self.calls += 1
        rows = [self.by_id[i] for i in ids]
        return {"documents": [self.docs[r] for r in rows], "metadatas": [self.metas[r] for r in rows]}


class CosineRetriever:
    """Stands in for FaissRetriever(embeddings, embedding_fn): exact cosine, best first, (ids, scores)."""

    def __init__(self, embeddings, embedding_fn):
        x = np.asarray(embeddings, np.float32)
        self.x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
        self.fn = embedding_fn

    def invoke(self, querys, k):
        q = np.asarray([self.fn.embed_query(s) for s in querys], np.float32)
        q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-30)
        sim = q @ self.x.T
        kk = min(k, self.x.shape[0])
        ids = np.full((len(querys), k), -1, np.int64)
        sc = np.full((len(querys), k), -np.finfo(np.float32).max, np.float32)
        for i in range(len(querys)):
            o = np.lexsort((np.arange(sim.shape[1]), -sim[i]))[:kk]
            ids[i, :kk], sc[i, :kk] = o, sim[i, o]
        return ids, sc


class Emb:
    def __init__(self, table):
        self.table = table

    def embed_query(self, text):
        return self.table[text]


class Bm25:
    def __init__(self, order, scores):
        self.order, self.scores = order, scores

    def invoke(self, query, k):
        return self.order[:k], self.scores[:k]


def _world(seed, n=240, d=24, n_titles=12):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, d)).astype(np.float32)
    # documents are chains of neighbouring chunks whose embeddings drift slowly: neighbours score high together
    for i in range(1, n):
        if i % 8:
            base[i] = 0.93 * base[i - 1] + 0.37 * base[i]
    metas, docs = [], []
    for i in range(n):
        first, last = i % 8 == 0, i % 8 == 7
        md = {"doc_id": f"d{i}", "prev_chunk_id": "" if first else f"d{i - 1}", "next_chunk_id": "" if last else f"d{i + 1}",
              "title_summary": f"title {i // (n // n_titles)}\nline"}
        if i % 5 == 0 or i % 5 == 1:            # two-row bundles scattered through the corpus
            md["bundle_id"] = f"b{i // 5}"
        if i % 31 == 0:
            md["bundle_id"] = None              # explicit null: treated as "no bundle"
        if i == 77:
            md["next_chunk_id"] = "missing"    # dangling neighbour id
        metas.append(md)
        docs.append(f"text of chunk {i}")
    titles = [f"title {t}\nline" for t in range(n_titles)] + ["title nobody has"]
    t_emb = rng.standard_normal((len(titles), d)).astype(np.float32)
    table = {}
    queries = []
    for j, anchor in enumerate((3, 42, 100, 77, 199, 238)):
        anchor %= n
        qv = base[anchor] + 0.05 * rng.standard_normal(d).astype(np.float32)
        table[f"q{j}"] = qv.tolist()
        table[f"h{j}"] = (base[(anchor + 2) % n] + 0.05 * rng.standard_normal(d).astype(np.float32)).tolist()
        queries.append((f"q{j}", [f"h{j}"] if j % 2 else []))
    chroma, ts = Store(docs, metas, base.tolist()), Store(titles, [None] * len(titles), t_emb.tolist())
    bm_order = rng.permutation(n).tolist()
    bm = Bm25(bm_order, np.sort(rng.random(n).astype(np.float32))[::-1])
    return chroma, ts, Emb(table), bm, queries, metas, titles


@pytest.mark.parametrize("expand", [False, True])
@pytest.mark.parametrize("prefetch", [False, True])
def test_invoke_matches_literal_restatement(expand, prefetch):
    from veritasfi_amd.ensemble import EnsembleRetriever
    chroma, ts, emb, bm, queries, metas, titles = _world(0)
    er = EnsembleRetriever("unused_dir", chroma, ts, 6, emb, faiss_ts_k=2, bm25_k=5, enable_expand=expand,
                           bm25_retriever=bm, retriever_cls=CosineRetriever, prefetch_documents=prefetch)
    dense, ts_dense = CosineRetriever(chroma.embs, emb), CosineRetriever(ts.embs, emb)
    ref_store = Store(chroma.docs, chroma.metas, chroma.embs)   # the restatement's own store: call counts stay apart
    grew = 0
    for q, hyde in queries:
        want = RE.gather(q, hyde, chunk_metadata=metas, title_summaries=titles,
                         store_get=lambda ids: ref_store.get(ids=ids, include=["documents", "metadatas"]),
                         dense=lambda t, k: tuple(a.tolist() for a in dense.invoke(t, k)),
                         ts_dense=lambda t, k: tuple(a.tolist() for a in ts_dense.invoke(t, k)),
                         bm25=bm.invoke, faiss_k=6, faiss_ts_k=2, bm25_k=5, enable_expand=expand)
        got = er.invoke(q, hyde)
        assert got == want
        assert {c["retriever"] for c in got} >= {"FAISS", "BM25"}
        assert all(isinstance(c["score"], float) for c in got)
        sizes = {}
        for c in got:
            sizes[c["bundle_id"]] = sizes.get(c["bundle_id"], 0) + 1
        grew += sum(1 for v in sizes.values() if v > 2)
        assert list(sizes) == list(range(len(sizes)))     # bundle ids count up in emission order
    if expand:
        assert grew > 0, "the corpus was built so that neighbour expansion fires"
    assert (er._documents is not None) == prefetch
    assert (chroma.calls == 0) == prefetch                 # prefetch: no per-bundle store round trips at all


def test_constructor_defaults_and_empty_cases():
    from veritasfi_amd.ensemble import EnsembleRetriever
    chroma, ts, emb, bm, queries, metas, titles = _world(1, n=40, n_titles=4)
    # ragManager.py:112 call shape.  The BM25 branch is on by default (bm25_k = k): without the host application's
    # BM25Retriever on the import path the constructor must fail, never drop a third of the recall silently
    with pytest.raises(ImportError, match="bm25_k"):
        EnsembleRetriever("dir", chroma, ts, 3, emb, retriever_cls=CosineRetriever)
    # ... and with it importable (ensembleRetriever.py:13,37) the unchanged call builds BM25Retriever(bm25_dir) itself
    import sys
    import types
    seen = []
    host = types.ModuleType("bm25Retriever")
    host.BM25Retriever = lambda d: (seen.append(d), bm)[1]
    sys.modules["bm25Retriever"] = host
    try:
        er = EnsembleRetriever("some/bm25_dir", chroma, ts, 3, emb, retriever_cls=CosineRetriever)
    finally:
        del sys.modules["bm25Retriever"]
    assert seen == ["some/bm25_dir"] and er.bm25_retriever is bm
    assert (er.faiss_k, er.faiss_ts_k, er.bm25_k, er.enable_expand) == (3, 3, 3, False)
    assert {c["retriever"] for q, h in queries for c in er.invoke(q, h)} == {"FAISS", "Title Summary", "BM25"}
    er = EnsembleRetriever("dir", chroma, ts, 3, emb, bm25_k=0, retriever_cls=CosineRetriever)   # explicitly off
    assert (er.faiss_k, er.faiss_ts_k, er.bm25_k) == (3, 3, 0) and er.bm25_retriever is None
    got = er.invoke(queries[0][0], [])
    assert got and all(c["retriever"] in ("FAISS", "Title Summary") for c in got)
    er0 = EnsembleRetriever("dir", chroma, ts, 0, emb, retriever_cls=CosineRetriever)   # k = 0: every branch off
    assert er0.invoke(queries[0][0], []) == []
    # corpus smaller than the 2048-deep search: padded ids (-1) are skipped, nothing is emitted twice
    rows = [c["metadata"]["doc_id"] for c in got]
    assert len(rows) == len(set(rows))


# The GPU end-to-end case compares with the REAL reference's output: tests/test_control_flow_golden.py
# (test_g5_gpu_ensemble_matches_reference_invoke).


def test_stage_hooks_carry_the_reference_names():
    """veritasfi_amd.set_profiler(p): EnsembleRetriever.invoke brackets "retrieve" and one stage per branch and reports the
    "retrieved_chunks" metric -- the names /root/reference/src/utils/ensembleRetriever.py:50,63,135,138,185,188,229,231 use -- and
    rank_chunk brackets "rerank" (vllmChatService.py:31) with its two device legs inside; without a profiler nothing is recorded, a
    skipped stage is left to the host, a stage that raises is still ended."""
    from datetime import datetime
    import veritasfi_amd as vf
    from veritasfi_amd import stages
    from veritasfi_amd.rank import rank_chunk
    import veritasfi_amd.rank as rank_mod

    class Rec:
        def __init__(self): self.log, self.metrics = [], {}
        def start(self, name): self.log.append(("start", name))
        def end(self, name): self.log.append(("end", name))
        def add_metric(self, name, value): self.metrics.setdefault(name, []).append(value)

    from veritasfi_amd.ensemble import EnsembleRetriever
    chroma, ts, emb, bm, queries, metas, titles = _world(2, n=40, n_titles=4)
    er = EnsembleRetriever("unused_dir", chroma, ts, 3, emb, faiss_ts_k=2, bm25_k=2, bm25_retriever=bm, retriever_cls=CosineRetriever)
    assert vf.get_profiler() is None
    base = er.invoke("q0", [])
    rec = Rec()
    prev = vf.set_profiler(rec)
    try:
        assert prev is None
        out = er.invoke("q0", [])
        assert out == base
        names = [n for kind, n in rec.log if kind == "start"]
        assert names == ["retrieve", "retrieve_faiss", "retrieve_faiss_ts", "retrieve_bm25"]
        assert rec.log[0] == ("start", "retrieve") and rec.log[-1] == ("end", "retrieve")
        assert rec.metrics["retrieved_chunks"] == [len(out)]
        # rank_chunk: the device legs replaced by host stand-ins (this is a CPU test of the brackets, not of the arithmetic)
        class RR:
            def compute_score(self, pairs, batch_size=8): return [float(len(p[1])) for p in pairs]
        chunks = [{"page_content": "x" * (i + 1), "bundle_id": i, "metadata": {"date_published": "2024-01-0%d" % (i + 1)}} for i in range(4)]
        saved = (rank_mod.fuse_and_rank, rank_mod.compute_similarity_mtx)
        rank_mod.fuse_and_rank = lambda a, b, dev=0: (None, list(np.argsort(-(np.asarray(a) + np.asarray(b)), kind="stable")))
        rank_mod.compute_similarity_mtx = lambda emb, texts, dev=0, as_torch=True, **kw: np.eye(len(texts), dtype=np.float32)
        try:
            rec.log.clear()
            got = rank_chunk(chunks, "q", datetime(2024, 1, 5), RR(), None, chunk_topk=2)
            assert got == [2, 3]
            assert rec.log == [("start", "rerank"), ("start", "rerank_score"), ("end", "rerank_score"), ("start", "rerank_similarity"),
                               ("end", "rerank_similarity"), ("end", "rerank")]
            vf.set_profiler(rec, skip=("rerank",))            # the host's own decorator keeps "rerank"
            rec.log.clear()
            rank_chunk(chunks, "q", datetime(2024, 1, 5), RR(), None, chunk_topk=2)
            assert ("start", "rerank") not in rec.log and ("start", "rerank_score") in rec.log
            class Boom:
                def compute_score(self, pairs, batch_size=8): raise RuntimeError("device lost")
            vf.set_profiler(rec)
            rec.log.clear()
            with pytest.raises(RuntimeError):
                rank_chunk(chunks, "q", datetime(2024, 1, 5), Boom(), None, chunk_topk=2)
            assert rec.log[-2:] == [("end", "rerank_score"), ("end", "rerank")]
        finally:
            rank_mod.fuse_and_rank, rank_mod.compute_similarity_mtx = saved
        with pytest.raises(TypeError):
            vf.set_profiler(object())
    finally:
        vf.set_profiler(None)
    rec.log.clear()
    er.invoke("q0", [])
    assert rec.log == []
    # the recorder that ships: per-thread timers, the reference's profile_data shape
    t = vf.StageTimer()
    vf.set_profiler(t)
    try:
        er.invoke("q0", [])
        er.invoke("q0", [])
    finally:
        vf.set_profiler(None)
    assert t.profile_data["retrieve"]["calls"] == 2 and len(t.profile_data["retrieve_faiss"]["execution_times"]) == 2
    assert t.metrics["retrieved_chunks"] == [len(base)] * 2 and t.summary()["retrieve"]["p50_ms"] >= t.summary()["retrieve_faiss"]["p50_ms"]
