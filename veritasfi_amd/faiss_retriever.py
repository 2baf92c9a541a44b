"""FaissRetriever -- drop-in for ``src/utils/faissRetriever.py:8-38`` backed by the HIP index.

Same constructor and ``invoke`` signature and return order as the reference class, so
``EnsembleRetriever`` (``src/utils/ensembleRetriever.py:40,43,66,139``) works unchanged when its
``from .faissRetriever import FaissRetriever`` is pointed here.
"""
from __future__ import annotations

import logging

import numpy as np

from .index import DenseIndex

logger = logging.getLogger(__name__)


class FaissRetriever:
    """Exact cosine retriever; the corpus lives in HBM, search runs as hand-written gfx950 kernels."""

    def __init__(self, embeddings, embedding_fn, device_id: int = 0, device_ids=None):
        # reference :13-24: np.array(embeddings) -> astype('float32') -> normalize_L2 -> IndexFlatIP.add
        self.embeddings = embedding_fn
        embeddings = np.array(embeddings)
        if embeddings.ndim != 2:
            raise ValueError("embeddings must be a 2-D array-like [n, d]")
        dimension = embeddings.shape[1]
        x = embeddings if embeddings.dtype == np.float16 else embeddings.astype("float32")
        # device_ids=[0..7]: the corpus is row-sharded over those GPUs behind the same handle (one process, no torchrun)
        self.index = DenseIndex(x, device_id=device_id, device_ids=device_ids)
        logger.info(f"Building HIP dense index with {len(embeddings)} vectors of dimension {dimension}")

    def invoke(self, querys: list, k: int):
        """(I, D) = ids and cosine scores, best first, shape [len(querys), k]; -1 / -FLT_MAX pad a corpus with fewer than k rows
        (reference :28-38: embed each string, fp32, L2-normalise, IndexFlatIP.search).  An embedder with a batched
        ``embed_queries`` -- ours has one -- gets ONE forward for all strings; any other is called per string, as upstream does."""
        texts = list(querys)
        embed_many = getattr(self.embeddings, "embed_queries", None)
        vectors = embed_many(texts) if embed_many is not None else [self.embeddings.embed_query(t) for t in texts]
        q = np.asarray(vectors, dtype=np.float32).reshape(len(texts), -1)
        ids, scores = self.index.search(q, k)      # normalisation happens on the device (k_prep_queries), canonical order
        return ids, scores
