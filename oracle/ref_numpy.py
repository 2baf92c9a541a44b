"""NumPy restatement of the reference's literal CPU path (TEST INFRASTRUCTURE ONLY).

Follows, line by line:

* ``sklearn.metrics.pairwise.cosine_similarity`` as called at
  ``experiments/retriever/step3_mul.py:243,275`` and ``continuous_retrieval.py:162``:
  ``normalize(X)``, ``normalize(Y)`` (row L2 norm via ``sqrt(einsum('ij,ij->i'))``, zero norms
  replaced by 1, elementwise division) followed by ``Xn @ Yn.T``.
* the per-row selection ``np.argsort(sim)[-k:][::-1]`` / ``np.argsort(sim)[::-1]`` for ``k == -1``
  (``step3_mul.py:245-249,278-283``, ``continuous_retrieval.py:164``).
* ``FaissRetriever`` arithmetic (``src/utils/faissRetriever.py:14-24,33-38``): normalise once at
  build, normalise the queries, inner product, best-k first, ``(indices, distances)`` order,
  ``k > N`` padded with ``-1`` / ``-FLT_MAX``.
* ``EnsembleRetriever.compute_similarity_mtx`` (``src/utils/ensembleRetriever.py:275-279``).
* ``ChatManager.rank_chunk`` score fusion (``src/utils/vllmManager.py:443-457``).

The matmul here goes through whatever BLAS NumPy links, exactly like the reference, so its
low-order score bits are not canonical; ``oracle.canonical`` is the bit-exact comparator.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.finfo(np.float32).max


def _normalize_rows(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    norms = np.sqrt(np.einsum("ij,ij->i", x, x))
    norms[norms == 0.0] = 1.0
    return x / norms[:, None]


def cosine_similarity(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """sklearn.cosine_similarity for dense fp32/fp64 inputs (dtype preserved)."""
    return _normalize_rows(x) @ _normalize_rows(y).T


def argsort_topk(similarities: np.ndarray, top_k: int) -> np.ndarray:
    """The reference's index selection for one row (unstable on exact ties, as upstream)."""
    if top_k == -1:
        return np.argsort(similarities)[::-1]
    return np.argsort(similarities)[-top_k:][::-1]


def select_top_chunks_batch(evidence_embs: np.ndarray, chunks_emb: np.ndarray, top_k: int):
    """step3_mul.select_top_chunks_batch after the two get_embeddings calls.

    Returns a list of (indices int64, similarities fp32) per evidence row.
    """
    sim = cosine_similarity(evidence_embs, chunks_emb)
    out = []
    for row in sim:
        idx = argsort_topk(row, top_k)
        out.append((idx.astype(np.int64), row[idx]))
    return out


def faiss_flat_ip_search(corpus: np.ndarray, queries: np.ndarray, k: int):
    """FaissRetriever.__init__ + invoke arithmetic; returns (indices, distances)."""
    x = _normalize_rows(np.asarray(corpus).astype("float32"))
    q = _normalize_rows(np.asarray(queries).astype("float32"))
    sim = q @ x.T
    nq, n = sim.shape
    ids = np.full((nq, k), -1, dtype=np.int64)
    dist = np.full((nq, k), -FLT_MAX, dtype=np.float32)
    kk = min(k, n)
    for i in range(nq):
        order = np.lexsort((np.arange(n), -sim[i]))[:kk]  # score desc, lower id first
        ids[i, :kk] = order
        dist[i, :kk] = sim[i, order]
    return ids, dist


def similarity_matrix(embs: np.ndarray) -> np.ndarray:
    """ensembleRetriever.compute_similarity_mtx after embedding: normalize, E @ E.T."""
    e = _normalize_rows(np.asarray(embs, dtype=np.float32))
    return e @ e.T


def time_scores(delta_days: np.ndarray) -> np.ndarray:
    """vllmManager.rank_chunk :443-447: max(0, 1 - |days| / 365)."""
    return np.maximum(0.0, 1.0 - np.abs(np.asarray(delta_days, dtype=np.float64)) / 365.0)


def fuse_and_rank(rerank_scores, time_sc) -> np.ndarray:
    """vllmManager.rank_chunk :454-457: scores = rerank + time; argsort descending
    (torch.argsort(descending=True) is not stable either; ties are broken lower-index-first
    here and in the product)."""
    s = np.asarray(rerank_scores, dtype=np.float32) + np.asarray(time_sc, dtype=np.float32)
    return np.lexsort((np.arange(s.shape[0]), -s)).astype(np.int64)


def rank_chunk(bundle_ids, rerank_scores, time_sc, embeddings, chunk_topk, similar_threshhold=0.9):
    """vllmManager.rank_chunk :430-483 after the model calls: bundle_ids per chunk, the re-ranker's scores,
    time scores and the chunk embeddings -> selected bundle ids (reverse selection order), including the
    reference's quirk of indexing the chunk similarity matrix with bundle ids (:476)."""
    n = len(bundle_ids)
    bundle_map = {}
    for idx, b in enumerate(bundle_ids):
        bundle_map.setdefault(b, []).append(idx)
    ranked = fuse_and_rank(rerank_scores, time_sc).tolist()
    sim = similarity_matrix(np.asarray(embeddings, dtype=np.float32)) if n else np.zeros((0, 0), np.float32)
    selected, size = [], 0
    for idx in ranked:
        b = bundle_ids[idx]
        if b in selected or size + len(bundle_map[b]) > chunk_topk:
            continue
        if selected and np.any(sim[idx, selected] > similar_threshhold):
            continue
        selected.append(b)
        size += len(bundle_map[b])
    return selected[::-1]


def e4m3_table() -> np.ndarray:
    """All 256 OCP FP8 E4M3 codes -> float32 (1-4-3, bias 7, no infinities, S.1111.111 = NaN).  The storage format
    BASELINE.json config [4] names; restated from the OCP 8-bit floating point specification (e4m3, "fn" flavour) and
    pinned against torch.float8_e4m3fn in tests/test_oracle_golden.py."""
    t = np.empty(256, dtype=np.float32)
    for b in range(256):
        sign = -1.0 if b & 0x80 else 1.0
        e, m = (b >> 3) & 0xF, b & 7
        if e == 0:
            v = m * 2.0 ** -9
        elif e == 15 and m == 7:
            v = float("nan")
        else:
            v = (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = sign * v
    return t


def decode_e4m3(codes: np.ndarray) -> np.ndarray:
    """uint8 e4m3 codes -> float32 values (exactly representable in fp16 as well)."""
    return e4m3_table()[np.asarray(codes, dtype=np.uint8)]
