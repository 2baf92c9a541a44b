"""Similarity matrix + score fusion: the small dense pieces around the re-ranker.

* ``compute_similarity_mtx`` / ``compute_similarity`` -- ``src/utils/ensembleRetriever.py:265-281`` and
  ``:235-263``: embed every chunk, L2-normalise, ``E @ E.T``.  The reference embeds the n chunks one
  ``embed_query`` at a time (``:275``); an embedder with ``embed_documents`` gets ONE batched call.
* ``time_scores`` / ``fuse_and_rank`` -- ``src/utils/vllmManager.py:443-457`` (``rank_chunk``):
  ``max(0, 1 - |days| / 365)``, ``rerank + time``, descending order.
"""
from __future__ import annotations

from datetime import datetime

import numpy as np

from . import index as _index


def _embed_all(embedding_fn, chunks):
    if hasattr(embedding_fn, "embed_documents"):
        return np.asarray(embedding_fn.embed_documents(list(chunks)), dtype=np.float32)
    return np.asarray([embedding_fn.embed_query(c) for c in chunks], dtype=np.float32)


def compute_similarity_mtx(embedding_fn, chunks, device_id: int = 0, as_torch: bool = True, index=None, row_ids=None):
    """ensembleRetriever.py:265-281.  Returns an [n, n] tensor the caller indexes as
    ``similar_mtx[idx, selected_indices] > 0.9`` (vllmManager.py:476).

    Default: the reference's route -- embed the n chunk texts (ONE batched call here) and take the canonical cosine.
    ``index=`` + ``row_ids=`` (explicit opt-in): the chunks came out of that ``DenseIndex`` and ``row_ids[i]`` is chunk i's row
    -- their embeddings are already in HBM, so the matrix is ``index.cosine_matrix_rows(row_ids)`` and nothing is embedded
    again.  The same matrix whenever the corpus rows ARE ``embed_documents(chunk texts)`` and the embedder treats documents and
    queries alike (the reference embeds both with ``embed_query``); an embedder with a query instruction must keep the default."""
    if len(chunks) == 0:
        mtx = np.zeros((0, 0), dtype=np.float32)
    elif index is not None:
        if row_ids is None or len(row_ids) != len(chunks):
            raise ValueError("row_ids must name one index row per chunk")
        mtx = index.cosine_matrix_rows(row_ids)
    else:
        mtx = _index.cosine_matrix(_embed_all(embedding_fn, chunks), device_id)
    if as_torch:
        import torch
        return torch.from_numpy(mtx)
    return mtx


def compute_similarity(embedding_fn, chunks, selected_indices, candidate_index, device_id: int = 0, as_torch=True):
    """ensembleRetriever.py:235-263: cosine of chunk `candidate_index` against the selected chunks."""
    embs = _embed_all(embedding_fn, chunks)
    if len(selected_indices) == 0:
        sims = np.zeros((0,), dtype=np.float32)
    else:
        sims = _index.cosine_scores(embs[list(selected_indices)], embs[[candidate_index]], device_id)[:, 0]
    if as_torch:
        import torch
        return torch.from_numpy(np.ascontiguousarray(sims))
    return sims


def time_scores(query_time: datetime, dates_published) -> np.ndarray:
    """vllmManager.py:443-447."""
    out = []
    for d in dates_published:
        days = abs((query_time - datetime.strptime(d, "%Y-%m-%d")).days)
        out.append(max(0, 1 - days / 365))
    return np.asarray(out, dtype=np.float32)


def fuse_and_rank(reranker_scores, time_sc, device_id: int = 0):
    """vllmManager.py:454-457: (scores, ranked_indices) with scores = rerank + time, best first."""
    scores, order = _index.fuse_rank(reranker_scores, time_sc, device_id)
    return scores, order.tolist()
