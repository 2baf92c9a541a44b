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

from . import stages
from .similarity import compute_similarity_mtx, fuse_and_rank


def rank_chunk(chunks, question: str, query_time: datetime, reranker, embedding_fn, chunk_topk: int,
               similar_threshhold: float = 0.9, reranker_lock=None, device_id: int = 0, similarity_index=None,
               row_id_key: str = "row_id"):
    """chunks: list of dicts with 'page_content', 'bundle_id', 'metadata'['date_published' = 'YYYY-MM-DD'].
    embedding_fn: an embedder (embed_documents / embed_query) -- or, as upstream passes it (vllmManager.py:430: the method's fourth
    argument is the RETRIEVER and :462 calls retriever.compute_similarity_mtx(texts)), any object with compute_similarity_mtx:
    this package's EnsembleRetriever then serves the texts it emitted from their corpus rows in HBM and embeds only the rest.
    similarity_index (explicit, see compute_similarity_mtx): the DenseIndex the chunks were retrieved from, each chunk carrying its
    row as chunk[row_id_key] -- the similarity matrix then comes from the corpus rows in HBM instead of re-embedding n texts.
    Stage brackets (stages.py; no-ops unless a profiler was set): "rerank" around the call -- the name upstream gives the function
    that wraps this method (vllmChatService.py:31) -- and "rerank_score" / "rerank_similarity" around its two device legs."""
    with stages.stage("rerank"):
        return _rank_chunk(chunks, question, query_time, reranker, embedding_fn, chunk_topk, similar_threshhold, reranker_lock,
                           device_id, similarity_index, row_id_key)


def _rank_chunk(chunks, question, query_time, reranker, embedding_fn, chunk_topk, similar_threshhold, reranker_lock, device_id,
                similarity_index, row_id_key):
    if not chunks:
        return []
    members = {}                                   # bundle id -> positions of its chunks
    for pos, chunk in enumerate(chunks):
        members.setdefault(chunk["bundle_id"], []).append(pos)
    texts = [chunk["page_content"] for chunk in chunks]
    # :443-447  max(0, 1 - |query date - chunk date| / 365)
    recency = [max(0, 1 - abs((query_time - datetime.strptime(chunk["metadata"]["date_published"], "%Y-%m-%d")).days) / 365)
               for chunk in chunks]
    with stages.stage("rerank_score"):
        with (reranker_lock if reranker_lock is not None else contextlib.nullcontext()):          # :450
            relevance = reranker.compute_score([[question, text] for text in texts], batch_size=8)
    _, order = fuse_and_rank(relevance, recency, device_id)                                       # :454-457
    with stages.stage("rerank_similarity"):
        if similarity_index is not None:
            sim = compute_similarity_mtx(embedding_fn, texts, device_id, as_torch=False, index=similarity_index,
                                         row_ids=[chunk[row_id_key] for chunk in chunks])
        elif hasattr(embedding_fn, "compute_similarity_mtx"):                                     # :462 retriever.compute_similarity_mtx
            sim = embedding_fn.compute_similarity_mtx(texts)
            sim = sim.numpy() if hasattr(sim, "numpy") else np.asarray(sim)
        else:
            sim = compute_similarity_mtx(embedding_fn, texts, device_id, as_torch=False)          # :462
    # :464-481 greedy pick, best fused score first: a bundle is taken whole if it is new, still fits chunk_topk, and its chunk is not
    # a near-duplicate (> threshold) of what was taken -- where "what was taken" is the list of BUNDLE ids used as COLUMN numbers of
    # the chunk-by-chunk matrix (the upstream quirk, :476; a bundle id >= n raises IndexError there and here)
    taken, size = [], 0
    for pos in order:
        bid = chunks[pos]["bundle_id"]
        n_members = len(members[bid])
        if bid in taken or size + n_members > chunk_topk:
            continue
        if taken and np.any(sim[pos, taken] > similar_threshhold):
            continue
        taken.append(bid)
        size += n_members
    return taken[::-1]                             # :483 reverse selection order
