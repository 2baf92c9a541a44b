"""One 768-wide index over text, table and figure chunks (BASELINE.json configs[3]).

The reference ingests text only (``src/load_data.py:98-128``: every chunk, tables included, goes through the text embedder's
``embed_documents``) and holds no image model; configs[3] adds a figure encoder (CLIP ViT) writing into the same 768-wide
matrix.  The rows of such a corpus live in TWO embedding spaces:

* the text embedder's space: text chunks and **tables as text** -- the reference's own route for tables (a table reaches
  Chroma as the text of its chunk); ``table-transformer``, which the config names, is a DETR-style *detector* that finds and
  segments tables upstream of ingest (boxes and classes, no pooled embedding): an upstream pre-processing step, not an embedder;
* CLIP's joint space: figure rows from the vision tower (``vf_vit_*``).

A cosine between vectors of different spaces means nothing, so a query is embedded by BOTH towers -- the text embedder and
the CLIP *text* tower (``vf_clip_text_*``) -- and each vector is scored against the rows of its own space only.  Rows are
grouped by modality into contiguous SEGMENTS of one matrix resident in HBM, one id space (the row number); each space is one
``DenseIndex`` borrowing its row range zero-copy (``id_offset`` = the range's first row), so every search is the exact,
oracle-identical ``vf_index_search`` and nothing is copied.
"""
from __future__ import annotations

import numpy as np

from .index import DenseIndex, _is_torch_tensor

TEXT_SPACE, CLIP_SPACE = "text", "clip"
DEFAULT_SPACES = {"text": TEXT_SPACE, "table": TEXT_SPACE, "figure": CLIP_SPACE}


class MixedModalIndex:
    def __init__(self, rows, segments: dict, spaces: dict | None = None, device_id: int = 0):
        """rows: [n, d] CUDA torch tensor (borrowed, the serving form) or ndarray (each space's range is copied to HBM).
        segments: modality -> (first row, end row); together they must tile [0, n) and the segments of one space must be
        adjacent.  spaces: modality -> space name (default: text and table rows in the text space, figure rows in CLIP's)."""
        spaces = dict(DEFAULT_SPACES if spaces is None else spaces)
        n = int(rows.shape[0])
        order = sorted(segments.items(), key=lambda kv: kv[1][0])
        at = 0
        for name, (lo, hi) in order:
            if name not in spaces:
                raise ValueError(f"segment {name!r} has no embedding space")
            if lo != at or hi < lo:
                raise ValueError("segments must tile [0, n) without gaps or overlaps")
            at = hi
        if at != n:
            raise ValueError("segments must tile [0, n) without gaps or overlaps")
        self.n, self.d = n, int(rows.shape[1])
        self.segments = {k: (int(a), int(b)) for k, (a, b) in segments.items()}
        self.space_of = spaces
        self._bounds = np.asarray([hi for _, (_, hi) in order], np.int64)
        self._names = [name for name, _ in order]
        self.ranges = {}
        for name, (lo, hi) in order:
            sp = spaces[name]
            if sp in self.ranges:
                if self.ranges[sp][1] != lo:
                    raise ValueError(f"the segments of space {sp!r} are not adjacent")
                self.ranges[sp] = (self.ranges[sp][0], hi)
            else:
                self.ranges[sp] = (lo, hi)
        self.indexes = {}
        for sp, (lo, hi) in self.ranges.items():
            if hi > lo:
                part = rows[lo:hi]
                self.indexes[sp] = DenseIndex(part, device_id=device_id, id_offset=lo) if _is_torch_tensor(part) \
                    else DenseIndex(np.ascontiguousarray(part), device_id=device_id, id_offset=lo)

    def search(self, vectors: dict, k: int) -> dict:
        """vectors: space -> [nq, d] query vectors of THAT space's tower.  Returns space -> (ids int64 [nq, k] global row
        numbers, scores fp32 [nq, k]), each exactly what ``vf_index_search`` over the space's rows returns."""
        out = {}
        for sp, q in vectors.items():
            if sp not in self.ranges:
                raise KeyError(f"no rows in space {sp!r}")
            q = np.ascontiguousarray(q, dtype=np.float32)
            if sp in self.indexes:
                out[sp] = self.indexes[sp].search(q, k)
            else:
                out[sp] = (np.full((q.shape[0], k), -1, np.int64), np.full((q.shape[0], k), -np.finfo(np.float32).max, np.float32))
        return out

    def modality_of(self, ids):
        """Segment name of every id (``None`` for the -1 padding)."""
        ids = np.asarray(ids)
        seg = np.searchsorted(self._bounds, ids, side="right")
        flat = [self._names[s] if 0 <= i < self.n else None for i, s in zip(ids.ravel().tolist(), seg.ravel().tolist())]
        return np.asarray(flat, dtype=object).reshape(ids.shape)

    def close(self):
        for ix in self.indexes.values():
            ix.close()
        self.indexes = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class MixedModalRetriever:
    """``invoke(querys, k)`` over a ``MixedModalIndex``: every query string is embedded by the text embedder AND by the CLIP
    text tower (both expose ``embed_documents``, the surface of src/utils/ragManager.py:50), each vector searches its own
    space, and the hits come back per space as ``(indices, distances)`` in ``FaissRetriever.invoke``'s order
    (src/utils/faissRetriever.py:28-38)."""

    def __init__(self, index: MixedModalIndex, text_embedder, clip_text_embedder):
        self.index = index
        self.embedders = {TEXT_SPACE: text_embedder, CLIP_SPACE: clip_text_embedder}

    def invoke(self, querys, k: int) -> dict:
        querys = list(querys)
        vecs = {sp: np.asarray(e.embed_documents(querys) if len(querys) != 1 else [e.embed_query(querys[0])], np.float32)
                for sp, e in self.embedders.items() if sp in self.index.ranges and e is not None}
        return self.index.search(vecs, k)
