"""Corpus file (.vfc) writer / reader -- the on-disk form of the embedding matrix.

Stands where the reference re-reads every embedding from Chroma into Python lists on each start
(``src/utils/ensembleRetriever.py:39-43`` -> ``src/utils/faissRetriever.py:14``): the embed loop
(``src/load_data.py:120-128``) appends its batches here once, and each rank's ``DenseIndex.from_file`` streams
only its row shard from disk into HBM (``vf_index_create_from_file``).  Layout: include/veritasfi_hip.h.
"""
from __future__ import annotations

import ctypes
import os
import struct

import numpy as np

from . import _ffi

MAGIC = b"VFCORPUS"
_HDR = struct.Struct("<8sIIQII32x")
_DT = {np.dtype(np.float32): _ffi.VF_DTYPE_F32, np.dtype(np.float16): _ffi.VF_DTYPE_F16, np.dtype(np.uint8): _ffi.VF_DTYPE_FP8_E4M3}
_NP = {v: k for k, v in _DT.items()}


class CorpusWriter:
    """Append row batches, then close (writes the header last so that a partial file never validates)."""

    def __init__(self, path: str, d: int, dtype=np.float16, e4m3: bool = False):
        self.path, self.d, self.n = path, int(d), 0
        self.dtype = np.dtype(np.uint8) if e4m3 else np.dtype(dtype)
        if self.dtype not in _DT:
            raise TypeError(f"unsupported corpus dtype {self.dtype}")
        self._ids = []
        self._f = open(path, "wb")
        self._f.write(b"\0" * _HDR.size)

    def append(self, rows, ids=None) -> None:
        rows = np.ascontiguousarray(np.asarray(rows), dtype=self.dtype) if self.dtype != np.uint8 else np.ascontiguousarray(rows)
        if rows.dtype != self.dtype or rows.ndim != 2 or rows.shape[1] != self.d:
            raise ValueError(f"batch must be [m, {self.d}] of {self.dtype}")
        if (ids is None) != (not self._ids) and self.n:
            raise ValueError("either every batch carries ids or none does")
        self._f.write(rows.tobytes())
        if ids is not None:
            ids = np.asarray(ids, dtype=np.int64)
            if ids.shape != (rows.shape[0],):
                raise ValueError("ids must be one int64 per row")
            self._ids.append(ids)
        self.n += rows.shape[0]

    def abort(self) -> None:
        """Leave on an error: the header is NOT written (the file keeps its zeroed first 64 bytes, which no reader
        accepts) and the partial file is removed."""
        if self._f is None:
            return
        self._f.close()
        self._f = None
        try:
            os.remove(self.path)
        except OSError:
            pass

    def close(self) -> None:
        if self._f is None:
            return
        if self._ids:
            self._f.write(np.concatenate(self._ids).tobytes())
        self._f.seek(0)
        self._f.write(_HDR.pack(MAGIC, 1, _DT[self.dtype], self.n, self.d, 1 if self._ids else 0))
        self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *a):
        if exc_type is not None:
            self.abort()
        else:
            self.close()


def write(path: str, rows, ids=None, e4m3: bool = False) -> None:
    rows = np.asarray(rows)
    with CorpusWriter(path, rows.shape[1], rows.dtype, e4m3=e4m3) as w:
        w.append(rows, ids)


def info(path: str) -> dict:
    """Header fields through the library's own validator (vf_corpus_file_info)."""
    n, d, dt, has = _ffi.c_i64(0), _ffi.c_i32(0), _ffi.c_i32(0), _ffi.c_i32(0)
    _ffi.check(_ffi.lib().vf_corpus_file_info(path.encode(), ctypes.byref(n), ctypes.byref(d), ctypes.byref(dt),
                                              ctypes.byref(has)), "vf_corpus_file_info")
    return {"n": n.value, "d": d.value, "dtype": dt.value, "has_ids": bool(has.value)}


def read_header(path: str) -> dict:
    """Pure-Python header read (no GPU library needed): for tools and CPU tests."""
    with open(path, "rb") as f:
        raw = f.read(_HDR.size)
    if len(raw) != _HDR.size:
        raise ValueError("corpus file too short")
    magic, version, dt, n, d, flags = _HDR.unpack(raw)
    if magic != MAGIC or version != 1 or dt not in _NP or d == 0:
        raise ValueError("not a version-1 corpus file")
    return {"n": n, "d": d, "dtype": dt, "has_ids": bool(flags & 1)}


def rows_memmap(path: str):
    h = read_header(path)
    return np.memmap(path, mode="r", dtype=_NP[h["dtype"]], offset=_HDR.size, shape=(h["n"], h["d"]))


def external_ids(path: str):
    """int64[n] external ids (row -> caller's id), or None when the file carries none."""
    h = read_header(path)
    if not h["has_ids"]:
        return None
    off = _HDR.size + h["n"] * h["d"] * _NP[h["dtype"]].itemsize
    return np.memmap(path, mode="r", dtype=np.int64, offset=off, shape=(h["n"],))


def embed_to_file(path: str, texts, embedder, batch_size: int = 100, dtype=np.float16, ids=None, on_batch=None) -> int:
    """The embed loop of ``import_collection_from_dir`` (``src/load_data.py:120-128,151``: batches of 100 texts through
    ``embed_documents``) writing a corpus file instead of one Chroma insert per batch.  ``embedder`` is anything with
    ``embed_documents(list[str]) -> list[list[float]]`` (``HipEmbeddings``, or the reference's ``HuggingFaceEmbeddings``).
    Returns the number of rows written."""
    texts = list(texts)
    if ids is not None and len(ids) != len(texts):
        raise ValueError("one id per text")
    w = None
    try:
        for i in range(0, len(texts), batch_size):
            vecs = np.asarray(embedder.embed_documents(texts[i:i + batch_size]), dtype=np.float32)
            if w is None:
                w = CorpusWriter(path, vecs.shape[1], dtype)
            w.append(vecs.astype(dtype), None if ids is None else np.asarray(ids[i:i + batch_size], dtype=np.int64))
            if on_batch is not None:
                on_batch(i + vecs.shape[0], len(texts))
        if w is None:
            raise ValueError("no texts to embed")
        n = w.n
    except BaseException:
        if w is not None:
            w.abort()  # a failed embed loop must not leave a truncated file that validates
        raise
    w.close()
    return n
