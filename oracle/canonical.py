"""ctypes binding of ``libvf_oracle.so`` (TEST INFRASTRUCTURE ONLY; see ``vf_oracle.c``).

Canonical-order restatement of the reference's cosine + top-k
(``experiments/retriever/step3_mul.py:233-289``, ``src/utils/faissRetriever.py:14-38``).
Bit-exact comparator for the HIP path: ids, rank order and score bits.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvf_oracle.so")
_lib = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_u16p = ctypes.POINTER(ctypes.c_uint16)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "vf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.vf_oracle_dot16.restype = ctypes.c_float
        L.vf_oracle_dot16.argtypes = [_f32p, _f32p, ctypes.c_int]
        L.vf_oracle_row_norms_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.vf_oracle_normalize_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.vf_oracle_cosine_f32.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.vf_oracle_topk_row.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _i64p, _f32p]
        L.vf_oracle_search_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int64, _i64p, _f32p]
        L.vf_oracle_search_f16.argtypes = [_u16p, ctypes.c_int64, ctypes.c_int, _f32p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int64, _i64p, _f32p]
        L.vf_oracle_merge_topk.argtypes = [_i64p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64p, _f32p]
        L.vf_oracle_num_threads.restype = ctypes.c_int
        L.vf_oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def dot16(a, b) -> np.float32:
    a, pa = _f32(a)
    b, pb = _f32(b)
    assert a.shape == b.shape and a.ndim == 1
    return np.float32(lib().vf_oracle_dot16(pa, pb, a.shape[0]))


def row_norms(x) -> np.ndarray:
    x, px = _f32(x)
    out = np.empty(x.shape[0], dtype=np.float32)
    _check(lib().vf_oracle_row_norms_f32(px, x.shape[0], x.shape[1], out.ctypes.data_as(_f32p)), "row_norms")
    return out


def normalize(x) -> np.ndarray:
    x, px = _f32(x)
    out = np.empty_like(x)
    _check(lib().vf_oracle_normalize_f32(px, x.shape[0], x.shape[1], out.ctypes.data_as(_f32p)), "normalize")
    return out


def cosine(a, b) -> np.ndarray:
    """Dense canonical cosine matrix [na, nb]."""
    a, pa = _f32(a)
    b, pb = _f32(b)
    assert a.shape[1] == b.shape[1]
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    _check(lib().vf_oracle_cosine_f32(pa, a.shape[0], pb, b.shape[0], a.shape[1], out.ctypes.data_as(_f32p)),
           "cosine")
    return out


def topk_row(scores, k: int):
    s, ps = _f32(scores)
    ids = np.empty(k, dtype=np.int64)
    out = np.empty(k, dtype=np.float32)
    _check(lib().vf_oracle_topk_row(ps, s.shape[0], k, ids.ctypes.data_as(_i64p), out.ctypes.data_as(_f32p)),
           "topk_row")
    return ids, out


def search(corpus, queries, k: int, id_offset: int = 0):
    """Exact cosine top-k; corpus fp32 or fp16 ndarray [n,d]; returns (ids int64 [nq,k], scores fp32 [nq,k])
    -- ids first, like FaissRetriever.invoke (faissRetriever.py:38)."""
    q, pq = _f32(queries)
    nq = q.shape[0]
    ids = np.empty((nq, k), dtype=np.int64)
    sc = np.empty((nq, k), dtype=np.float32)
    c = np.ascontiguousarray(corpus)
    n, d = c.shape
    assert q.shape[1] == d
    if c.dtype == np.float16:
        rc = lib().vf_oracle_search_f16(c.view(np.uint16).ctypes.data_as(_u16p), n, d, pq, nq, k, id_offset,
                                        ids.ctypes.data_as(_i64p), sc.ctypes.data_as(_f32p))
    else:
        c = np.ascontiguousarray(c, dtype=np.float32)
        rc = lib().vf_oracle_search_f32(c.ctypes.data_as(_f32p), n, d, pq, nq, k, id_offset,
                                        ids.ctypes.data_as(_i64p), sc.ctypes.data_as(_f32p))
    _check(rc, "search")
    return ids, sc


def merge_topk(ids_parts, score_parts, k: int):
    """Merge per-shard [G, nq, k] results into [nq, k]."""
    ids_in = np.ascontiguousarray(ids_parts, dtype=np.int64)
    sc_in = np.ascontiguousarray(score_parts, dtype=np.float32)
    g, nq, kk = ids_in.shape
    assert kk == k and sc_in.shape == ids_in.shape
    ids = np.empty((nq, k), dtype=np.int64)
    sc = np.empty((nq, k), dtype=np.float32)
    _check(lib().vf_oracle_merge_topk(ids_in.ctypes.data_as(_i64p), sc_in.ctypes.data_as(_f32p), g, nq, k,
                                      ids.ctypes.data_as(_i64p), sc.ctypes.data_as(_f32p)), "merge")
    return ids, sc


def num_threads() -> int:
    return int(lib().vf_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().vf_oracle_set_num_threads(int(n))
