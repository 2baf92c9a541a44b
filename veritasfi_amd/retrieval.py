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


def last_token_pool(last_hidden_states, attention_mask):
    """step3_mul.py:181-188, including its quirk: ``[:, -1]`` only when EVERY row ends in a 1."""
    import torch
    left_padding = (attention_mask[:, -1].sum() == attention_mask.shape[0])
    if left_padding:
        return last_hidden_states[:, -1]
    sequence_lengths = attention_mask.sum(dim=1) - 1
    batch_size = last_hidden_states.shape[0]
    return last_hidden_states[torch.arange(batch_size, device=last_hidden_states.device), sequence_lengths]


def get_embeddings(texts, model, tokenizer, device, batch_size=32, pooling="last_token", max_length=None):
    """step3_mul.py:191-209 (``pooling="last_token"``, max_length 4096) and
    continuous_retrieval.py:127-152 (``pooling="mean"``: unmasked ``mean(dim=1)``, max_length 512).
    ``model`` is any callable with the HF signature (our HIP encoder wrapper or an HF module)."""
    import torch
    if not texts:
        return np.array([])
    if max_length is None:
        max_length = 4096 if pooling == "last_token" else 512
    all_embeddings = []
    for i in range(0, len(texts), batch_size):
        batch_texts = texts[i:i + batch_size]
        inputs = tokenizer(batch_texts, padding=True, truncation=True, return_tensors="pt", max_length=max_length)
        if hasattr(model, "pooled"):  # this package's HipModel: forward + pooling in one GPU call
            all_embeddings.append(np.asarray(model.pooled(pooling, **inputs), dtype=np.float32))
            continue
        inputs = {k: v.to(device) for k, v in inputs.items()}
        with torch.no_grad():
            outputs = model(**inputs)
            if pooling == "last_token":
                embeddings = last_token_pool(outputs.last_hidden_state, inputs["attention_mask"])
            else:
                embeddings = outputs.last_hidden_state.mean(dim=1)
            all_embeddings.append(embeddings.float().cpu().numpy())
    return np.vstack(all_embeddings)


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
