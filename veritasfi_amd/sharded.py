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
    """local_index: object with ``search_device(q, k) -> (ids, scores)`` returning GLOBAL ids
    (a DenseIndex built with id_offset = shard start).  merge_fn: ([G,nq,k] ids, [G,nq,k] scores, k)
    -> ([nq,k], [nq,k]); defaults to the HIP merge kernel."""

    def __init__(self, local_index, group=None, merge_fn=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.local = local_index
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._packed = merge_fn is None  # default: packed single-collective exchange + HIP merge
        self._bufs = {}
        self.merge_fn = merge_fn

    def search(self, queries, k: int):
        """queries: [nq, d] tensor on this rank's device, identical on every rank."""
        import torch
        if self.world == 1:
            return self.local.search_device(queries, k)
        nq = int(queries.shape[0])
        if self._packed:
            # production path: the shard writes (ids, scores) into one packed blob, ONE all-gather moves every
            # rank's blob (nq*k*12 bytes each: 77 KB at nq=64, k=100), the HIP merge reads the parts in place
            key = (nq, k, queries.device)
            if self._bufs.get("key") != key:
                from .index import packed_result_buffer
                blob, ids, sc = packed_result_buffer(nq, k, queries.device)
                self._bufs = {"key": key, "blob": blob, "ids": ids, "sc": sc,
                              "all": torch.empty(self.world * nq * k * 12, dtype=torch.uint8, device=queries.device)}
            b = self._bufs
            self.local.search_device(queries, k, b["ids"], b["sc"])
            self.dist.all_gather_into_tensor(b["all"], b["blob"], group=self.group)
            from .index import merge_topk_packed_device
            return merge_topk_packed_device(b["all"], self.world, nq, k)
        ids, scores = self.local.search_device(queries, k)
        # outputs are the rank-order concatenation along dim 0 (the layout every backend accepts);
        # rank order == ascending id range, which the merge relies on
        all_ids = torch.empty((self.world * nq, k), dtype=ids.dtype, device=ids.device)
        all_sc = torch.empty((self.world * nq, k), dtype=scores.dtype, device=scores.device)
        self.dist.all_gather_into_tensor(all_ids, ids.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(all_sc, scores.contiguous(), group=self.group)
        return self.merge_fn(all_ids.view(self.world, nq, k), all_sc.view(self.world, nq, k), k)
