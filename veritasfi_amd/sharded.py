"""Row-sharded retrieval across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU retrieval (its only data parallelism is a model replica per worker
process, ``experiments/retriever/step3_mul.py:405-452``); SURVEY.md 8(e) defines this path: rank g
holds rows ``[g*ceil(N/G), (g+1)*ceil(N/G))``, every rank searches its shard for the same query batch,
the per-shard ``(score, global id)`` top-k lists are exchanged with ONE all-gather
(``B*k*12`` bytes per rank: 77 KB at B=64, k=100) and merged on every rank by ``vf_merge_topk_device``.
Scores are canonical (independent of the sharding), so the merged result is bit-identical to the
single-GPU result.

``torch.distributed`` is the transport only (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests,
which inject an oracle-backed local index -- the product never runs on CPU).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous row block of `rank` (SURVEY 8e partitioning)."""
    per = -(-n // world)
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class ShardedRetriever:
    """local_index: object with ``search_device(q, k, out_ids, out_scores) -> (ids, scores)`` returning GLOBAL ids
    (a DenseIndex built with id_offset = shard start).

    Default exchange (``merge_fn is None``): the shard writes its result into ONE packed blob
    (``index.packed_result_buffer``), ONE ``all_gather_into_tensor`` moves every rank's blob and
    ``packed_merge_fn(all_blobs, world, nq, k)`` merges the parts in place -- by default the HIP kernel behind
    ``vf_merge_topk_packed_device``.  ``merge_fn`` (two typed all-gathers + ``merge_fn([G,nq,k] ids, [G,nq,k]
    scores, k)``) is kept for callers that bring their own merge."""

    def __init__(self, local_index, group=None, merge_fn=None, packed_merge_fn=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.local = local_index
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._packed = merge_fn is None  # default: packed single-collective exchange + HIP merge
        self._bufs = {}
        self.merge_fn = merge_fn
        if packed_merge_fn is None:
            from .index import merge_topk_packed_device as packed_merge_fn
        self.packed_merge_fn = packed_merge_fn
        # RCCL ("nccl") moves device buffers directly over xGMI.  Any other backend (gloo: the CPU tests and the
        # one-GPU rehearsal of world > 1) gets the blob through a host copy -- same bytes, same single collective.
        self._direct = (not dist.is_initialized()) or dist.get_backend(group) == "nccl"

    def _all_gather_blob(self, out, blob):
        if self._direct or not blob.is_cuda:
            self.dist.all_gather_into_tensor(out, blob, group=self.group)
            return
        import torch
        h_out = torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(h_out, blob.cpu(), group=self.group)
        out.copy_(h_out)

    def search(self, queries, k: int):
        """queries: [nq, d] tensor on this rank's device, identical on every rank."""
        import torch
        if self.world == 1:
            return self.local.search_device(queries, k)
        nq = int(queries.shape[0])
        if self._packed:
            key = (nq, k, queries.device)
            if self._bufs.get("key") != key:
                from .index import packed_part_bytes, packed_result_buffer
                blob, ids, sc = packed_result_buffer(nq, k, queries.device)
                self._bufs = {"key": key, "blob": blob, "ids": ids, "sc": sc,
                              "all": torch.empty(self.world * packed_part_bytes(nq, k), dtype=torch.uint8,
                                                 device=queries.device)}
            b = self._bufs
            self.local.search_device(queries, k, b["ids"], b["sc"])
            self._all_gather_blob(b["all"], b["blob"])
            return self.packed_merge_fn(b["all"], self.world, nq, k)
        ids, scores = self.local.search_device(queries, k)
        # outputs are the rank-order concatenation along dim 0 (the layout every backend accepts);
        # rank order == ascending id range, which the merge relies on
        all_ids = torch.empty((self.world * nq, k), dtype=ids.dtype, device=ids.device)
        all_sc = torch.empty((self.world * nq, k), dtype=scores.dtype, device=scores.device)
        self.dist.all_gather_into_tensor(all_ids, ids.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(all_sc, scores.contiguous(), group=self.group)
        return self.merge_fn(all_ids.view(self.world, nq, k), all_sc.view(self.world, nq, k), k)


class ShardedScorer:
    """Data-parallel re-ranking / embedding over the ranks of one node ("replicas only ... split the 100 pairs
    across GPUs and all-gather 100 floats").  Every rank holds a replica of the model; rank r scores the contiguous block
    ``shard_bounds(n, world, r)`` of the items and ONE all-gather returns all ``n x width`` values to every rank, in item
    order.  ``score_fn(lo, hi) -> array [hi - lo] or [hi - lo, width]`` is the local model call (e.g.
    ``lambda lo, hi: encoder.forward(ids[lo:hi], mask[lo:hi])``); it is not called for an empty block.

    The reference scores its candidates on one device behind a lock (``src/utils/vllmManager.py:450-452``); this is the
    multi-GPU form of that call, nothing else."""

    def __init__(self, score_fn, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.score_fn = score_fn
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, n: int, width: int = 1) -> np.ndarray:
        import torch
        lo, hi = shard_bounds(n, self.world, self.rank)
        local = np.zeros((0, width), dtype=np.float32)
        if hi > lo:
            local = np.asarray(self.score_fn(lo, hi), dtype=np.float32).reshape(hi - lo, width)
        if self.world == 1:
            return local[:, 0] if width == 1 else local
        per = -(-n // self.world)                        # every rank sends `per` rows (the last blocks are padded)
        send = torch.zeros((per, width), dtype=torch.float32)
        send[:hi - lo] = torch.from_numpy(local)
        on_dev = self.device is not None and self.dist.get_backend(self.group) == "nccl"
        if on_dev:
            send = send.to(self.device)
        out = torch.empty((self.world * per, width), dtype=torch.float32, device=send.device)
        self.dist.all_gather_into_tensor(out, send, group=self.group)
        res = out[:n].cpu().numpy()                      # block r starts at r * per: rank order is item order
        return res[:, 0] if width == 1 else res
