"""Experiment-path functions with the reference's names and signatures.

Mirrors ``experiments/retriever/step3_mul.py:181-289`` (``last_token_pool``, ``get_embeddings``,
``select_top_chunks``, ``select_top_chunks_batch``) and ``experiments/retriever/continuous_retrieval.py:127-167``
(mean-pool ``get_embeddings``, ``select_top_chunks`` returning chunks only).  The cosine + top-k --
``sklearn.cosine_similarity`` + ``np.argsort`` upstream -- runs on the GPU through ``vf_index_*``;
results come back in the product's declared order (descending, lower index first on exact ties,
where upstream's order is whatever ``np.argsort`` leaves).
"""
from __future__ import annotations

import numpy as np

from .index import DenseIndex


def _host_array(x) -> np.ndarray:
    """ndarray view of a tokenizer / model output (numpy array, or a CPU torch tensor as HF-style callables return)."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def last_token_pool(last_hidden_states, attention_mask):
    """The hidden state of each sequence's LAST token, with the reference's rule for telling the padding side
    (``experiments/retriever/step3_mul.py:181-188``): when the final column of the mask is 1 in EVERY row the batch counts as
    left-padded and column -1 is taken for all rows; otherwise row i yields position ``mask[i].sum() - 1`` (so a right-padded batch
    in which one row happens to be full length still takes the second branch -- and a mixed batch follows the same arithmetic as
    upstream, quirk included).  Works on numpy arrays and on torch tensors alike (indexing only); the type of the input comes back."""
    rows = attention_mask.shape[0]
    if int(attention_mask[:, -1].sum()) == rows:
        return last_hidden_states[:, -1]
    ends = attention_mask.sum(1) - 1
    return last_hidden_states[list(range(rows)), ends]


def get_embeddings(texts, model, tokenizer, device=None, batch_size=32, pooling="last_token", max_length=None):
    """``get_embeddings(texts, model, tokenizer, device, batch_size)`` of both experiment scripts -- ``step3_mul.py:191-209``
    (``pooling="last_token"``, truncation at 4096) and ``continuous_retrieval.py:127-152`` (``pooling="mean"``: the UNMASKED
    ``mean(dim=1)`` over whatever the tokenizer padded to, truncation at 512) -- returning float32 ``[n, hidden]``.

    ``model`` is one of this package's HF-signature wrappers (``HipModel`` / ``HipDecoderModel``): the forward runs on the GPU.  With
    ``model.pooled`` the pooling happens there too (one call per batch, nothing but ``[b, hidden]`` comes back); a callable that only
    offers ``model(**inputs).last_hidden_state`` -- the reference's generic shape -- has its states pooled here on the host with the
    rule above.  ``device`` is accepted for signature compatibility; the handles know their device."""
    texts = list(texts)
    if not texts:
        return np.array([])
    limit = max_length if max_length is not None else (4096 if pooling == "last_token" else 512)
    if pooling not in ("last_token", "mean"):
        raise ValueError("pooling must be 'last_token' or 'mean'")
    on_device = getattr(model, "pooled", None)
    pieces = []
    for start in range(0, len(texts), batch_size):
        enc = tokenizer(texts[start:start + batch_size], padding=True, truncation=True, return_tensors="pt", max_length=limit)
        if on_device is not None:
            pieces.append(np.asarray(on_device(pooling, **enc), dtype=np.float32))
            continue
        states = _host_array(model(**enc).last_hidden_state).astype(np.float32, copy=False)
        if pooling == "mean":
            pieces.append(states.mean(axis=1, dtype=np.float32))
        else:
            pieces.append(np.asarray(last_token_pool(states, _host_array(enc["attention_mask"])), dtype=np.float32))
    return np.vstack(pieces)


def top_chunks_from_embeddings(evidence_embs, chunks_emb, top_k: int, device_id: int = 0):
    """cosine_similarity(E, C) + per-row argsort top-k (step3_mul.py:275-283) on the GPU.
    Returns (ids int64 [E, kk], sims float32 [E, kk]) with kk = C when top_k == -1."""
    chunks_emb = np.asarray(chunks_emb)
    kk = chunks_emb.shape[0] if top_k == -1 else int(top_k)
    kk = min(kk, chunks_emb.shape[0])  # np.argsort(s)[-k:] cannot return more than C entries
    with DenseIndex(chunks_emb, device_id=device_id) as index:
        return index.search(np.asarray(evidence_embs, dtype=np.float32), kk)


def select_top_chunks(evidence, query_chunks, model, tokenizer, device, top_k=3, batch_size=32,
                      pooling="last_token", return_similarities=True, device_id: int = 0):
    """step3_mul.py:233-253 -> (top_chunks, top_similarities); with ``return_similarities=False``
    the continuous_retrieval.py:154-167 form -> top_chunks."""
    if not query_chunks:
        return ([], []) if return_similarities else []
    evidence_emb = get_embeddings([evidence], model, tokenizer, device, batch_size=1, pooling=pooling)
    chunks_emb = get_embeddings(query_chunks, model, tokenizer, device, batch_size=batch_size, pooling=pooling)
    ids, sims = top_chunks_from_embeddings(evidence_emb, chunks_emb, top_k, device_id)
    top_chunks = [query_chunks[i] for i in ids[0]]
    if not return_similarities:
        return top_chunks
    return top_chunks, [np.float32(s) for s in sims[0]]


def select_top_chunks_batch(evidence_list, query_chunks, model, tokenizer, device, top_k=3, batch_size=32,
                            pooling="last_token", device_id: int = 0):
    """step3_mul.py:255-289 -> [(top_chunks, top_similarities), ...] per evidence."""
    if not evidence_list or not query_chunks:
        return [([], [])] * len(evidence_list)
    evidence_embs = get_embeddings(evidence_list, model, tokenizer, device, batch_size=batch_size, pooling=pooling)
    chunks_emb = get_embeddings(query_chunks, model, tokenizer, device, batch_size=batch_size, pooling=pooling)
    ids, sims = top_chunks_from_embeddings(evidence_embs, chunks_emb, top_k, device_id)
    results = []
    for row_ids, row_sims in zip(ids, sims):
        results.append(([query_chunks[i] for i in row_ids], [np.float32(s) for s in row_sims]))
    return results
