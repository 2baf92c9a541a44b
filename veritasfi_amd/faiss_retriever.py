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

    def __init__(self, embeddings, embedding_fn, device_id: int = 0, device_ids=None, corpus_dtype: str = "f32"):
        # reference :13-24: np.array(embeddings) -> astype('float32') -> normalize_L2 -> IndexFlatIP.add
        self.embeddings = embedding_fn
        embeddings = np.array(embeddings)
        if embeddings.ndim != 2:
            raise ValueError("embeddings must be a 2-D array-like [n, d]")
        dimension = embeddings.shape[1]
        # corpus_dtype (the optional configuration key of SURVEY.md section 5): how the rows are HELD in HBM -- "f32" as faiss holds
        # them (fp16 input stays fp16), "f16" halves the bytes the scan reads, "fp8" (OCP e4m3) quarters them (BASELINE configs[4]).
        # Scores are the canonical cosine of the STORED values either way.
        # device_ids=[0..7]: the corpus is row-sharded over those GPUs behind the same handle (one process, no torchrun)
        # rows_as_given: the index holds the embeddings' own values (nothing was rounded to a narrower type), so a cosine taken from
        # the rows is the cosine of the embeddings (EnsembleRetriever.compute_similarity_mtx serves known texts from them)
        self.rows_as_given = corpus_dtype == "f32" or (corpus_dtype == "f16" and embeddings.dtype == np.float16)
        if corpus_dtype == "f32":
            x = embeddings if embeddings.dtype == np.float16 else embeddings.astype("float32")
            self.index = DenseIndex(x, device_id=device_id, device_ids=device_ids)
        elif corpus_dtype == "f16":
            self.index = DenseIndex(embeddings.astype(np.float16), device_id=device_id, device_ids=device_ids)
        elif corpus_dtype == "fp8":
            import torch
            # torch's cast to e4m3 does not saturate (|x| > 448 becomes a NaN code) and flushes small magnitudes (the components of a
            # unit vector of 768+ dimensions sit around the subnormal edge 2^-6).  A cosine does not change under a positive per-row
            # scale, so every row is scaled by the POWER OF TWO that puts its largest magnitude into [224, 448]: exact in fp32 (the
            # rounding of in-range values is what it was), nothing can overflow, and the small components keep their three bits
            x = np.ascontiguousarray(embeddings.astype(np.float32))
            if not np.isfinite(x).all():
                raise ValueError("corpus_dtype='fp8': the embeddings hold non-finite values")
            peak = np.abs(x).max(axis=1, keepdims=True).astype(np.float64)
            x = x * np.exp2(np.floor(np.log2(448.0 / np.where(peak > 0, peak, 448.0)))).astype(np.float32)
            np.clip(x, -448.0, 448.0, out=x)
            codes = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
            if ((codes & 0x7F) == 0x7F).any():
                raise ValueError("corpus_dtype='fp8': the cast produced NaN codes")
            self.index = DenseIndex.from_e4m3(codes, device_id=device_id, device_ids=device_ids)
        else:
            raise ValueError(f"corpus_dtype {corpus_dtype!r}: one of f32, f16, fp8")
        logger.info(f"Building HIP dense index with {len(embeddings)} vectors of dimension {dimension}")

    @classmethod
    def from_index(cls, index: DenseIndex, embedding_fn, rows_as_given: bool = True):
        """The same retriever over an index that already exists -- ``DenseIndex.from_file`` (the corpus file the embed loop wrote,
        corpus_file.py: no Chroma round trip at start-up, ensembleRetriever.py:39-43), a device tensor, a sharded group.  Use it as
        ``EnsembleRetriever(..., retriever_cls=lambda _embeddings, fn: FaissRetriever.from_index(ix, fn))``.
        rows_as_given: the index holds the embedder's own values (see __init__)."""
        self = cls.__new__(cls)
        self.embeddings, self.index, self.rows_as_given = embedding_fn, index, bool(rows_as_given)
        return self

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
