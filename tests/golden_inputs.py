"""Deterministic inputs for the golden fixtures in tests/golden/.

The fixtures store only (seed parameters, sha256 of the generated input bytes, the REFERENCE's
outputs); the inputs are regenerated here and their checksum is verified, so a drift in NumPy's
generator fails loudly instead of silently comparing different data.  Used by tools/gen_golden.py
(which runs the real reference on these arrays) and by the tests.  No reference code here.
"""
import hashlib

import numpy as np

# (E evidences, C chunks, d, k) -- k == -1 means "all, sorted" (step3_mul.py:245-246)
G2_CASES = [
    (3, 500, 64, 5),
    (8, 2000, 768, 100),
    (64, 20000, 768, 100),
    (5, 700, 1024, -1),
    (16, 5000, 1024, 5),
    (4, 3000, 384, 10),
    (1, 150, 768, 3),
]


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def g1_inputs():
    rng = np.random.default_rng(0)
    chunks = rng.standard_normal((1000, 768)).astype(np.float32)
    evid = rng.standard_normal((1, 768)).astype(np.float32)
    return chunks, evid


def g2_inputs(ci: int):
    E, C, d, k = G2_CASES[ci]
    rng = np.random.default_rng(100 + ci)
    chunks = rng.standard_normal((C, d)).astype(np.float32)
    evid = rng.standard_normal((E, d)).astype(np.float32)
    return chunks, evid, k


def g3_inputs():
    """C2-like distribution: N(0,1) corpus rounded to fp16, fp32 queries."""
    corpus = np.random.default_rng(1234).standard_normal((4096, 768)).astype(np.float32).astype(np.float16)
    queries = np.random.default_rng(4321).standard_normal((16, 768)).astype(np.float32)
    return corpus, queries


def g4_inputs():
    """Exact ties: duplicated rows, a positively scaled copy, a zero row."""
    rng = np.random.default_rng(7)
    base = rng.standard_normal((40, 64)).astype(np.float32)
    chunks = np.vstack([base, base[[5, 10, 20]], 2.0 * base[[5]], np.zeros((1, 64), np.float32)])
    evid = base[[5]] + 0.01 * rng.standard_normal((1, 64)).astype(np.float32)
    tie_groups = [[5, 40, 43], [10, 41], [20, 42]]
    return chunks, evid, tie_groups


# ---------------------------------------------------------------------------------------------------
# G5-G7: control-flow fixtures (EnsembleRetriever.invoke, ChatManager.rank_chunk, get_inputs).  The
# generators below build the INPUTS only (a Chroma-shaped store, injected retrievers, chunk dicts, a
# word-level tokenizer on HF's own PreTrainedTokenizer); tools/gen_golden.py hands them to the real
# reference classes and records what those return.
# ---------------------------------------------------------------------------------------------------
FLT_MAX = float(np.finfo(np.float32).max)


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
    """Stands in for FaissRetriever(embeddings, embedding_fn) where faiss is absent: exact cosine in NumPy,
    best first, returns (ids int64 [nq,k], scores fp32 [nq,k]) padded with (-1, -FLT_MAX) like faiss."""

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
        sc = np.full((len(querys), k), -FLT_MAX, np.float32)
        for i in range(len(querys)):
            o = np.lexsort((np.arange(sim.shape[1]), -sim[i]))[:kk]
            ids[i, :kk], sc[i, :kk] = o, sim[i, o]
        return ids, sc


class TableEmbeddings:
    """embed_query / embed_documents by table lookup (text -> vector)."""

    def __init__(self, table):
        self.table = table

    def embed_query(self, text):
        return self.table[text]

    def embed_documents(self, texts):
        return [self.table[t] for t in texts]


class ListBM25:
    """BM25Retriever.invoke(query, k) -> (ids, scores) from a fixed ranking (bm25Retriever.py:50-87 surface)."""

    def __init__(self, order, scores):
        self.order, self.scores = order, scores

    def invoke(self, query, k):
        return self.order[:k], self.scores[:k]


G5_CASES = [  # (seed, n, k, faiss_k, faiss_ts_k, bm25_k, enable_expand)
    (0, 240, 6, None, 2, 5, False),
    (0, 240, 6, None, 2, 5, True),
    (2, 240, 6, None, 2, 4, True),
    (3, 96, 5, 7, 3, 0, True),       # no BM25 branch, explicit faiss_k
    (4, 160, 4, 0, 4, 6, False),     # no dense branch: title summaries + BM25 only
    (5, 2400, 8, None, 1, 3, True),  # corpus larger than the 2048-deep search: no padding in the score map
]


def g5_world(seed, n=240, d=24, n_titles=12):
    """A corpus of 8-chunk documents whose neighbouring chunks drift slowly (so neighbour expansion fires),
    two-row bundles scattered through it, explicit-null bundle ids, a dangling neighbour id, titles nobody has."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, d)).astype(np.float32)
    for i in range(1, n):
        if i % 8:
            base[i] = 0.93 * base[i - 1] + 0.37 * base[i]
    metas, docs = [], []
    for i in range(n):
        first, last = i % 8 == 0, i % 8 == 7
        md = {"doc_id": f"d{i}", "prev_chunk_id": "" if first else f"d{i - 1}", "next_chunk_id": "" if last else f"d{i + 1}",
              "title_summary": f"title {i // (n // n_titles)}\nline", "date_published": f"2024-{1 + i % 12:02d}-{1 + i % 28:02d}"}
        if i % 5 == 0 or i % 5 == 1:
            md["bundle_id"] = f"b{i // 5}"
        if i % 31 == 0:
            md["bundle_id"] = None
        if i == 77:
            md["next_chunk_id"] = "missing"
        metas.append(md)
        docs.append(f"text of chunk {i}")
    titles = [f"title {t}\nline" for t in range(n_titles)] + ["title nobody has"]
    t_emb = rng.standard_normal((len(titles), d)).astype(np.float32)
    table, queries = {}, []
    for j, anchor in enumerate((3, 42, 100, 77, 199, 238)):
        anchor %= n
        table[f"q{j}"] = (base[anchor] + 0.05 * rng.standard_normal(d).astype(np.float32)).tolist()
        table[f"h{j}"] = (base[(anchor + 2) % n] + 0.05 * rng.standard_normal(d).astype(np.float32)).tolist()
        table[f"g{j}"] = (base[(anchor + 9) % n] + 0.05 * rng.standard_normal(d).astype(np.float32)).tolist()
        queries.append((f"q{j}", [[], [f"h{j}"], [f"h{j}", f"g{j}"]][j % 3]))
    bm_order = rng.permutation(n).tolist()
    bm_scores = np.sort(rng.random(n).astype(np.float32))[::-1]
    return {"docs": docs, "metas": metas, "embs": base, "titles": titles, "t_embs": t_emb, "table": table,
            "queries": queries, "bm_order": bm_order, "bm_scores": bm_scores}


G6_CASES = [  # (seed, n chunks, chunk_topk, kind)
    (0, 24, 5, "plain"),
    (1, 60, 8, "plain"),
    (2, 155, 10, "plain"),          # stress_test.py:153 size
    (3, 40, 6, "near_duplicates"),   # clusters of chunks with cosine > 0.9: the dedupe rule decides
    (8, 90, 9, "near_duplicates"),
    (4, 30, 4, "tied_scores"),       # exact score ties inside one bundle (the order among them cannot matter)
    (5, 12, 5, "bundle_ids_beyond_n"),   # bundle ids >= n: the :476 quirk indexes the matrix out of range
    (6, 1, 3, "plain"),
    (7, 20, 2, "big_bundles"),       # bundles larger than chunk_topk are skipped
]


def g6_inputs(ci):
    """chunk dicts (page_content, metadata.date_published, bundle_id), re-ranker scores by text, embeddings by text,
    the query time."""
    seed, n, topk, kind = G6_CASES[ci]
    rng = np.random.default_rng(600 + seed)
    d = 32
    emb = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "near_duplicates":        # three clusters of near-identical chunks: a third of all pairs sit above 0.9
        centre = rng.standard_normal((3, d)).astype(np.float32)
        emb = centre[np.arange(n) % 3] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
    rr = (3.0 * rng.standard_normal(n)).astype(np.float32)
    sizes, bundle, b = [], [], 0
    while len(bundle) < n:
        sz = int(rng.integers(1, 7 if kind == "big_bundles" else 4))
        bundle.extend([b] * sz)
        b += 1
    bundle = bundle[:n]
    if kind == "bundle_ids_beyond_n":
        bundle = [v + 3 * n for v in bundle]
    if kind == "tied_scores":
        for i in range(1, n):
            if bundle[i] == bundle[i - 1]:
                rr[i] = rr[i - 1]
    chunks = []
    for i in range(n):
        date = f"{2023 + int(rng.integers(0, 3))}-{int(rng.integers(1, 13)):02d}-{int(rng.integers(1, 29)):02d}"
        if kind == "tied_scores" and i and bundle[i] == bundle[i - 1]:
            date = chunks[-1]["metadata"]["date_published"]
        chunks.append({"retriever": "FAISS", "score": 0.5, "page_content": f"chunk text {ci}/{i}",
                       "metadata": {"doc_id": f"d{i}", "date_published": date}, "bundle_id": int(bundle[i])})
    rr_by_text = {c["page_content"]: float(rr[i]) for i, c in enumerate(chunks)}
    emb_by_text = {c["page_content"]: emb[i].tolist() for i, c in enumerate(chunks)}
    return {"chunks": chunks, "rr": rr_by_text, "emb": emb_by_text, "chunk_topk": topk,
            "query_time": (2024, 6, 15, 10, 30, 0), "question": f"question {ci}"}


G7_WORDS = ["revenue", "lotus", "margin", "battery", "delivery", "2023", "guidance", "segment", "cash", "vehicle",
            "Given", "a", "query", "A", "and", "passage", "B,", "determine", "whether", "the", "contains", "an",
            "answer", "to", "by", "providing", "prediction", "of", "either", "'Yes'", "or", "'No'.", "A:", "B:", "Yes", "No"]
G7_CASES = [  # (seed, n pairs, max_length, padding_side)
    (0, 11, 96, "left"),
    (1, 8, 64, "right"),
    (2, 5, 1024, "left"),
    (3, 9, 40, "left"),      # tight: the query alone nearly fills max_length
]


def g7_pairs(ci):
    seed, n, max_length, side = G7_CASES[ci]
    rng = np.random.default_rng(700 + seed)
    sent = lambda m: " ".join(G7_WORDS[int(i)] for i in rng.integers(0, 10, m))
    pairs = [[sent(int(rng.integers(3, 12))), sent(int(rng.integers(5, 60)))] for _ in range(n - 2)]
    pairs.append([sent(max_length), sent(max_length + 20)])          # both truncations (3/4 query, only_second)
    pairs.append([sent(2), "line one\nline two " + sent(4)])         # separators inside the passage
    return pairs, max_length, side


def g7_tokenizer(padding_side="left"):
    """A word-level vocabulary on HF's own (pure Python) PreTrainedTokenizer, so that ``__call__``, ``prepare_for_model``
    and ``pad`` -- the three methods get_inputs calls -- are transformers' real implementations."""
    from transformers import PreTrainedTokenizer

    class WordTokenizer(PreTrainedTokenizer):
        def __init__(self, **kw):
            self._vocab = {"<pad>": 0, "<unk>": 1, "<bos>": 2, "<eos>": 3, "\n": 4}
            for w in G7_WORDS + ["line", "one", "two"]:
                self._vocab.setdefault(w, len(self._vocab))
            self._inv = {v: k for k, v in self._vocab.items()}
            super().__init__(pad_token="<pad>", unk_token="<unk>", bos_token="<bos>", eos_token="<eos>", **kw)

        @property
        def vocab_size(self):
            return len(self._vocab)

        def get_vocab(self):
            return dict(self._vocab)

        def _tokenize(self, text, **kw):
            return [w for w in text.replace("\n", " \n ").split(" ") if w != ""]

        def _convert_token_to_id(self, token):
            return self._vocab.get(token, 1)

        def _convert_id_to_token(self, index):
            return self._inv.get(index, "<unk>")

    return WordTokenizer(padding_side=padding_side)
