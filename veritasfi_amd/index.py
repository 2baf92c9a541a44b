"""DenseIndex -- Python handle over ``vf_index_*`` (exact cosine top-k on one MI355X).

Stands where ``faiss.IndexFlatIP`` + ``faiss.normalize_L2`` stand in the reference
(``src/utils/faissRetriever.py:18-24,35-37``).  NumPy in / NumPy out for the drop-in classes; torch
CUDA tensors in / out (zero copy, device pointers through the C ABI) for the serving and multi-GPU
paths.  PyTorch is plumbing here: device memory and streams only.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _dev_array(device_ids):
    ids = [int(x) for x in device_ids]
    if not ids:
        raise ValueError("device_ids must name at least one device")
    return (_ffi.c_i32 * len(ids))(*ids), ids


class DenseIndex:
    def __init__(self, rows, device_id: int = 0, id_offset: int = 0, device_ids=None):
        """rows: [n, d] float32 / float16 ndarray (copied to HBM) or a CUDA torch tensor (borrowed).
        FP8: a torch.float8_e4m3fn tensor (CPU or CUDA), or a uint8 ndarray / tensor of OCP e4m3 codes passed with
        ``DenseIndex.from_e4m3``; the bytes stay fp8 in HBM (scanned as fp8, converted exactly in registers).
        device_ids=[...]: ONE handle over several GPUs in this process (``vf_index_create_sharded``): host rows are
        split into contiguous blocks, one per listed device; every search method works on it unchanged, device
        buffers live on the home device ``device_ids[0]``."""
        L = _ffi.lib()
        self._h = _ffi.vp()
        self._keepalive = None
        self.device_id = int(device_id)
        self.device_ids = None
        e4m3 = getattr(self, "_e4m3", False)
        if device_ids is not None:
            if _is_torch_tensor(rows):
                import torch
                if rows.dtype == getattr(torch, "float8_e4m3fn", None):
                    rows, e4m3 = rows.view(torch.uint8), True
                rows = rows.cpu().numpy()   # the sharded constructor distributes HOST rows (DenseIndex.group adopts device shards)
            rows = np.asarray(rows)
            if rows.ndim != 2:
                raise ValueError("rows must be [n, d]")
            if e4m3:
                if rows.dtype != np.uint8:
                    raise TypeError("e4m3 rows must be uint8 codes")
                dt = _ffi.VF_DTYPE_FP8_E4M3
            elif rows.dtype == np.float16:
                dt = _ffi.VF_DTYPE_F16
            else:
                rows, dt = rows.astype(np.float32, copy=False), _ffi.VF_DTYPE_F32
            rows = np.ascontiguousarray(rows)
            arr, ids = _dev_array(device_ids)
            self.n, self.d, self.id_offset = int(rows.shape[0]), int(rows.shape[1]), 0
            self.device_id, self.device_ids = ids[0], ids
            _ffi.check(L.vf_index_create_sharded(ctypes.byref(self._h), rows.ctypes.data, self.n, self.d, dt, arr, len(ids)),
                       "vf_index_create_sharded")
            self._warn_if_staged()
            return
        if _is_torch_tensor(rows):
            import torch
            if rows.dtype == getattr(torch, "float8_e4m3fn", None):
                rows, e4m3 = rows.view(torch.uint8), True
            if not rows.is_cuda:
                rows = rows.cpu().numpy()
            else:
                if rows.dim() != 2 or not rows.is_contiguous():
                    raise ValueError("rows must be a contiguous [n, d] tensor")
                if e4m3 and rows.dtype == torch.uint8:
                    dt = _ffi.VF_DTYPE_FP8_E4M3
                elif rows.dtype == torch.float16:
                    dt = _ffi.VF_DTYPE_F16
                elif rows.dtype == torch.float32:
                    dt = _ffi.VF_DTYPE_F32
                else:
                    raise TypeError(f"unsupported corpus dtype {rows.dtype}")
                self.device_id = rows.device.index if rows.device.index is not None else self.device_id
                self._keepalive = rows
                self.n, self.d = int(rows.shape[0]), int(rows.shape[1])
                _ffi.check(L.vf_index_create_device(ctypes.byref(self._h), rows.data_ptr(), self.n, self.d, dt,
                                                    self.device_id, int(id_offset)), "vf_index_create_device")
                self.id_offset = int(id_offset)
                return
        rows = np.asarray(rows)
        if rows.ndim != 2:
            raise ValueError("rows must be [n, d]")
        if e4m3:
            if rows.dtype != np.uint8:
                raise TypeError("e4m3 rows must be uint8 codes")
            dt = _ffi.VF_DTYPE_FP8_E4M3
        elif rows.dtype == np.float16:
            dt = _ffi.VF_DTYPE_F16
        else:
            rows = rows.astype(np.float32, copy=False)  # faissRetriever.py:21  x = embeddings.astype('float32')
            dt = _ffi.VF_DTYPE_F32
        rows = np.ascontiguousarray(rows)
        self.n, self.d = int(rows.shape[0]), int(rows.shape[1])
        self.id_offset = int(id_offset)
        _ffi.check(L.vf_index_create(ctypes.byref(self._h), rows.ctypes.data, self.n, self.d, dt, self.device_id,
                                     self.id_offset), "vf_index_create")

    @classmethod
    def from_file(cls, path: str, rank: int = 0, world: int = 1, device_id: int = 0, device_ids=None):
        """Rows of this rank's shard of a corpus file (veritasfi_amd/corpus_file.py), streamed disk -> HBM by the
        library; returned ids are file row numbers (id_offset = the shard's first row).  device_ids=[...]: the whole
        file, one block per listed device, behind one handle (single-process multi-GPU)."""
        from .sharded import shard_bounds
        n, d, dt, has = _ffi.c_i64(0), _ffi.c_i32(0), _ffi.c_i32(0), _ffi.c_i32(0)
        L = _ffi.lib()
        _ffi.check(L.vf_corpus_file_info(path.encode(), ctypes.byref(n), ctypes.byref(d), ctypes.byref(dt),
                                         ctypes.byref(has)), "vf_corpus_file_info")
        self = cls.__new__(cls)
        self._h = _ffi.vp()
        self._keepalive = None
        self.device_ids = None
        if device_ids is not None:
            arr, ids = _dev_array(device_ids)
            self.device_id, self.device_ids, self.id_offset = ids[0], ids, 0
            self.n, self.d = int(n.value), int(d.value)
            _ffi.check(L.vf_index_create_sharded_from_file(ctypes.byref(self._h), path.encode(), arr, len(ids)),
                       "vf_index_create_sharded_from_file")
            self._warn_if_staged()
            return self
        lo, hi = shard_bounds(n.value, world, rank)
        self.device_id, self.id_offset = int(device_id), int(lo)
        self.n, self.d = int(hi - lo), int(d.value)
        _ffi.check(L.vf_index_create_from_file(ctypes.byref(self._h), path.encode(), lo, hi, self.device_id, lo),
                   "vf_index_create_from_file")
        return self

    @classmethod
    def group(cls, shards):
        """Adopt per-device indexes (contiguous row blocks in ascending order, each built with id_offset = its first
        row) as ONE handle (``vf_index_group``).  The group owns them: the given objects are emptied."""
        shards = list(shards)
        L = _ffi.lib()
        arr = (_ffi.vp * len(shards))(*[s._h for s in shards])
        self = cls.__new__(cls)
        self._h = _ffi.vp()
        _ffi.check(L.vf_index_group(ctypes.byref(self._h), arr, len(shards)), "vf_index_group")
        self._keepalive = [s._keepalive for s in shards]   # borrowed device rows stay alive with the group
        self.device_ids = [s.device_id for s in shards]
        self.device_id, self.id_offset = shards[0].device_id, shards[0].id_offset
        self.n, self.d = sum(s.n for s in shards), shards[0].d
        for s in shards:
            s._h = _ffi.vp()
            s._keepalive = None
        self._warn_if_staged()
        return self

    @classmethod
    def from_e4m3(cls, codes, device_id: int = 0, id_offset: int = 0, device_ids=None):
        """codes: [n, d] uint8 OCP-e4m3 bytes (ndarray, CPU or CUDA tensor)."""
        self = cls.__new__(cls)
        self._e4m3 = True
        self.__init__(codes, device_id=device_id, id_offset=id_offset, device_ids=device_ids)
        return self

    # -- host buffers ------------------------------------------------------------------------------
    def search(self, queries, k: int):
        """queries [nq, d] -> (ids int64 [nq, k], scores float32 [nq, k]); ids first, as
        FaissRetriever.invoke returns them (faissRetriever.py:38)."""
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"queries must be [nq, {self.d}], got {q.shape}")
        k = int(k)
        ids = np.empty((q.shape[0], k), dtype=np.int64)
        scores = np.empty((q.shape[0], k), dtype=np.float32)
        _ffi.check(_ffi.lib().vf_index_search(self._h, q.ctypes.data, q.shape[0], k, ids.ctypes.data,
                                              scores.ctypes.data), "vf_index_search")
        return ids, scores

    # -- device buffers (torch CUDA tensors) ---------------------------------------------------------
    def _dev_args(self, queries, k, out_ids, out_scores):
        import torch
        if not (queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()):
            raise ValueError("queries must be a contiguous float32 CUDA tensor")
        nq = int(queries.shape[0])
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=queries.device)
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
        stream = torch.cuda.current_stream(queries.device).cuda_stream
        return nq, out_ids, out_scores, stream

    def search_device(self, queries, k: int, out_ids=None, out_scores=None):
        nq, out_ids, out_scores, stream = self._dev_args(queries, int(k), out_ids, out_scores)
        _ffi.check(_ffi.lib().vf_index_search_device(self._h, queries.data_ptr(), nq, int(k), out_ids.data_ptr(),
                                                     out_scores.data_ptr(), stream), "vf_index_search_device")
        return out_ids, out_scores

    def search_begin(self, slot: int, queries, k: int, out_ids=None, out_scores=None):
        """Enqueue a batch into `slot` and return at once; results are valid after search_end(slot).
        The caller keeps `queries` alive until then."""
        nq, out_ids, out_scores, stream = self._dev_args(queries, int(k), out_ids, out_scores)
        _ffi.check(_ffi.lib().vf_index_search_begin(self._h, int(slot), queries.data_ptr(), nq, int(k),
                                                    out_ids.data_ptr(), out_scores.data_ptr(), stream),
                   "vf_index_search_begin")
        return out_ids, out_scores

    def search_end(self, slot: int):
        _ffi.check(_ffi.lib().vf_index_search_end(self._h, int(slot)), "vf_index_search_end")

    @property
    def slots(self) -> int:
        out = _ffi.c_i32(0)
        _ffi.check(_ffi.lib().vf_index_slots(self._h, ctypes.byref(out)), "vf_index_slots")
        return int(out.value)

    # -- misc --------------------------------------------------------------------------------------
    def shard_devices(self) -> list:
        """Devices of the shards behind this handle ([] for a plain single-device index)."""
        n = _ffi.c_i32(0)
        devs = (_ffi.c_i32 * 64)()
        _ffi.check(_ffi.lib().vf_index_shards(self._h, ctypes.byref(n), devs, 64), "vf_index_shards")
        return [int(devs[i]) for i in range(min(int(n.value), 64))]

    def peer_access(self) -> list:
        """Per shard: True when xGMI peer access between the home device and the shard's device is enabled in both
        directions (or they are the same device); False = that shard's query / result copies are staged through the host."""
        ok = (_ffi.c_i32 * 64)()
        missing = _ffi.c_i32(0)
        _ffi.check(_ffi.lib().vf_index_peer_access(self._h, ok, 64, ctypes.byref(missing)), "vf_index_peer_access")
        return [bool(ok[i]) for i in range(len(self.shard_devices()))]

    def _warn_if_staged(self):
        flags = self.peer_access()
        if flags and not all(flags):
            import warnings
            devs = self.shard_devices()
            warnings.warn("veritasfi_amd: no peer (xGMI) access between the home device and device(s) "
                          f"{[d for d, f in zip(devs, flags) if not f]}: the exchange of those shards is staged through "
                          "the host (correct, slower)", RuntimeWarning, stacklevel=3)

    def stats(self) -> dict:
        st = _ffi.SearchStats()
        _ffi.check(_ffi.lib().vf_index_stats(self._h, ctypes.byref(st)), "vf_index_stats")
        return st.as_dict()

    def profile(self) -> dict:
        """HIP-event totals since set_option("profile", 1): main k_scan ms / launches, pipeline ms."""
        ms, n, pipe, nbytes = ctypes.c_double(0), _ffi.c_i64(0), ctypes.c_double(0), _ffi.c_i64(0)
        _ffi.check(_ffi.lib().vf_index_profile(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(pipe),
                                               ctypes.byref(nbytes)), "vf_index_profile")
        span, nl = ctypes.c_double(0), _ffi.c_i64(0)
        _ffi.check(_ffi.lib().vf_index_profile_span(self._h, ctypes.byref(span), ctypes.byref(nl)), "vf_index_profile_span")
        return {"scan_ms_total": ms.value, "scan_launches": int(n.value), "pipeline_ms_total": pipe.value,
                "scan_bytes_per_launch": int(nbytes.value), "span_ms": span.value}

    def set_option(self, name: str, value: int) -> None:
        _ffi.check(_ffi.lib().vf_index_set_option(self._h, name.encode(), int(value)), "vf_index_set_option")

    def cosine_matrix_rows(self, ids, extra=None) -> np.ndarray:
        """[n, n] canonical cosine matrix of rows already in this index, picked by global id (``vf_cosine_matrix_rows``): the
        similarity matrix of retrieved chunks without re-embedding their texts.
        ``extra`` ([m, d] fp32): positions whose id is -1 take the next row of ``extra`` instead (``vf_cosine_matrix_rows_mixed``)
        -- texts the corpus does not hold, embedded by the caller."""
        ids = np.ascontiguousarray(ids, dtype=np.int64).ravel()
        out = np.empty((ids.size, ids.size), np.float32)
        if not ids.size:
            return out
        if extra is None or len(extra) == 0:
            _ffi.check(_ffi.lib().vf_cosine_matrix_rows(self._h, ids.ctypes.data, int(ids.size), out.ctypes.data),
                       "vf_cosine_matrix_rows")
            return out
        extra = np.ascontiguousarray(extra, dtype=np.float32)
        if extra.ndim != 2 or extra.shape[1] != self.d:
            raise ValueError(f"extra must be [m, {self.d}]")
        _ffi.check(_ffi.lib().vf_cosine_matrix_rows_mixed(self._h, ids.ctypes.data, int(ids.size), extra.ctypes.data,
                                                          int(extra.shape[0]), out.ctypes.data), "vf_cosine_matrix_rows_mixed")
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.lib().vf_index_destroy(self._h)
            self._h = _ffi.vp()
        self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def cosine_scores(a, b, device_id: int = 0) -> np.ndarray:
    """Canonical cosine matrix [na, nb] (step3_mul.py:275 ``cosine_similarity(E, C)``)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("cosine_scores: a [na, d], b [nb, d]")
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    _ffi.check(_ffi.lib().vf_cosine_scores(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], a.shape[1],
                                           out.ctypes.data, int(device_id)), "vf_cosine_scores")
    return out


def cosine_matrix(x, device_id: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    out = np.empty((x.shape[0], x.shape[0]), dtype=np.float32)
    _ffi.check(_ffi.lib().vf_cosine_matrix(x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data, int(device_id)),
               "vf_cosine_matrix")
    return out


def merge_topk_device(ids_parts, score_parts, k: int):
    """[G, nq, k] CUDA tensors (parts in ascending id-range order) -> ([nq, k], [nq, k])."""
    import torch
    g, nq, kk = ids_parts.shape
    assert kk == k and ids_parts.is_contiguous() and score_parts.is_contiguous()
    ids = torch.empty((nq, k), dtype=torch.int64, device=ids_parts.device)
    sc = torch.empty((nq, k), dtype=torch.float32, device=ids_parts.device)
    dev = ids_parts.device.index or 0
    _ffi.check(_ffi.lib().vf_merge_topk_device(ids_parts.data_ptr(), score_parts.data_ptr(), g, nq, k, ids.data_ptr(),
                                               sc.data_ptr(), dev, torch.cuda.current_stream(ids_parts.device).cuda_stream),
               "vf_merge_topk_device")
    return ids, sc


def packed_part_bytes(nq: int, k: int) -> int:
    """Bytes of one packed per-shard result: [ids nq*k int64][scores nq*k fp32], padded to a multiple of 16 so that
    every part of an all-gathered buffer keeps its int64 ids aligned (the library computes the same stride)."""
    return (nq * k * 12 + 15) // 16 * 16


def packed_result_buffer(nq: int, k: int, device):
    """One contiguous device blob [ids nq*k int64][scores nq*k fp32][pad to 16 B] plus the two typed views into it:
    a shard writes its search result through the views and ships the blob with a single all-gather."""
    import torch
    blob = torch.zeros(packed_part_bytes(nq, k), dtype=torch.uint8, device=device)
    ids = blob[: nq * k * 8].view(torch.int64).view(nq, k)
    scores = blob[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
    return blob, ids, scores


def merge_topk_packed_device(parts_blob, nparts: int, nq: int, k: int, out_ids=None, out_scores=None):
    """parts_blob: uint8 tensor of nparts packed results (rank order == ascending id ranges)."""
    import torch
    dev = parts_blob.device
    if out_ids is None:
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    if out_scores is None:
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().vf_merge_topk_packed_device(parts_blob.data_ptr(), nparts, nq, k, out_ids.data_ptr(),
                                                      out_scores.data_ptr(), dev.index or 0,
                                                      torch.cuda.current_stream(dev).cuda_stream),
               "vf_merge_topk_packed_device")
    return out_ids, out_scores


def fuse_rank(rerank_scores, time_scores, device_id: int = 0):
    a = np.ascontiguousarray(np.asarray(rerank_scores, dtype=np.float32))
    b = np.ascontiguousarray(np.asarray(time_scores, dtype=np.float32))
    assert a.shape == b.shape and a.ndim == 1
    out = np.empty_like(a)
    order = np.empty(a.shape[0], dtype=np.int64)
    _ffi.check(_ffi.lib().vf_fuse_rank(a.ctypes.data, b.ctypes.data, a.shape[0], out.ctypes.data, order.ctypes.data,
                                       int(device_id)), "vf_fuse_rank")
    return out, order
