"""rank_chunk -- the consumer of the re-ranker and the similarity matrix, with the reference's semantics.

Mirrors ``ChatManager.rank_chunk`` (``src/utils/vllmManager.py:430-483``) as a free function: what that
method reads from ``self`` (``reranker``, ``reranker_lock``, ``chunk_topk``, ``similar_threshhold``) and from
its ``retriever`` argument (``compute_similarity_mtx``) is passed in.  The arithmetic runs on the GPU
(``reranker.compute_score`` -> vf_reranker_score, ``fuse_and_rank`` -> vf_fuse_rank,
``compute_similarity_mtx`` -> vf_encoder_forward + vf_cosine_matrix); the greedy selection over a few dozen
chunks is host control flow, as upstream.

Faithful quirks (kept on purpose, they decide the output):
* ``selected_indices`` holds BUNDLE ids, and the similarity test indexes the chunk-by-chunk matrix with those
  bundle ids as column numbers (``similar_mtx[idx, selected_indices]``, ``:476``);
* the result is the selected bundle ids in REVERSE selection order (``:483``);
* exact score ties: lower index first (``torch.argsort`` leaves ties unspecified upstream).
"""
from __future__ import annotations

import contextlib
from datetime import datetime

import numpy as np

from .similarity import compute_similarity_mtx, fuse_and_rank


def rank_chunk(chunks, question: str, query_time: datetime, reranker, embedding_fn, chunk_topk: int,
               similar_threshhold: float = 0.9, reranker_lock=None, device_id: int = 0, similarity_index=None,
               row_id_key: str = "row_id"):
    """chunks: list of dicts with 'page_content', 'bundle_id', 'metadata'['date_published' = 'YYYY-MM-DD'].
    similarity_index (opt-in, see compute_similarity_mtx): the DenseIndex the chunks were retrieved from, each chunk carrying its
    row as chunk[row_id_key] -- the similarity matrix then comes from the corpus rows in HBM instead of re-embedding n texts."""
    bundle_map = {}
    for idx, chunk in enumerate(chunks):
        bundle_map.setdefault(chunk["bundle_id"], []).append(idx)
    pairs = [[question, chunk["page_content"]] for chunk in chunks]
    chunk_content_list = [chunk["page_content"] for chunk in chunks]
    time_scores = []
    for chunk in chunks:  # :443-447  max(0, 1 - |query date - chunk date| / 365)
        score = abs((query_time - datetime.strptime(chunk["metadata"]["date_published"], "%Y-%m-%d")).days)
        time_scores.append(max(0, 1 - score / 365))
    if not chunks:
        return []
    with (reranker_lock if reranker_lock is not None else contextlib.nullcontext()):  # :450
        reranker_scores = reranker.compute_score(pairs, batch_size=8)
    _, ranked_indices = fuse_and_rank(reranker_scores, time_scores, device_id)       # :454-457
    if similarity_index is not None:
        similar_mtx = compute_similarity_mtx(embedding_fn, chunk_content_list, device_id, as_torch=False, index=similarity_index,
                                             row_ids=[chunk[row_id_key] for chunk in chunks])
    else:
        similar_mtx = compute_similarity_mtx(embedding_fn, chunk_content_list, device_id, as_torch=False)  # :462
    selected_indices = []
    current_size = 0
    for idx in ranked_indices:                                                        # :464-481
        bundle_id = chunks[idx]["bundle_id"]
        bundle = bundle_map[bundle_id]
        if bundle_id in selected_indices or current_size + len(bundle) > chunk_topk:
            continue
        if selected_indices and np.any(similar_mtx[idx, selected_indices] > similar_threshhold):
            continue
        selected_indices.append(bundle_id)
        current_size += len(bundle)
    return selected_indices[::-1]
