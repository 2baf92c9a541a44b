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
        # reference :33-38: one embed_query per string, fp32, normalise, search, return (I, D)
        # (an embedder that offers a batched embed_queries -- ours does -- gets one forward instead of
        #  len(querys); any other embedder is called exactly as the reference calls it)
        if hasattr(self.embeddings, "embed_queries"):
            query_vec_list = self.embeddings.embed_queries(list(querys))
        else:
            query_vec_list = [self.embeddings.embed_query(q) for q in querys]
        query_vector = np.array(query_vec_list).astype("float32")
        if query_vector.ndim == 1:
            query_vector = query_vector.reshape(len(querys), -1)
        indices, distances = self.index.search(query_vector, k)
        return indices, distances
