"""Row-sharded episodic bank: one process per GPU, per-shard top-k, RCCL all-gather, merge.

SURVEY.md section 8e: the reference has no multi-GPU path; the bank shards naturally by rows and
recall needs exactly one exchange step.  Rank g owns global rows ``[g*rows_per_rank,
(g+1)*rows_per_rank)``.  A recall is

    local scan (aura_knn_search, idx_base = first global row)      -> (score, global_row)[nq, k]
    all_gather of nq*k*8 bytes per rank (64 KiB at nq=256, k=32)   -> [S, nq, k] on every rank
    aura_topk_merge                                                 -> global top-k on every rank

The message is tiny, so the collective is latency bound; RCCL's all_gather over xGMI with every
GPU directly linked to its 7 peers is a single hop.  ``torch.distributed`` (backend "nccl" = RCCL
on ROCm, "gloo" in the CPU tests) is the only plumbing used.

``all_gather_queries=True`` additionally gathers each rank's own query block first, which is the
serving layout used by ``bench.py --gpus N``: every rank brings nq queries, scans its shard for
all S*nq of them, and ends up with the global top-k of its own nq.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [begin, end) owned by ``rank`` (remainder spread over low ranks)."""
    base, rem = divmod(total_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class ShardedRecall:
    """Collective recall over a row-sharded bank.

    ``local_search(queries [n, D], k) -> (scores [n, k] fp32, rows [n, k] int32 GLOBAL rows)`` and
    ``merge(scores [S, n, k], rows [S, n, k], k) -> (scores [n, k], rows [n, k])`` are injected so
    that the same host logic runs on the HIP ops (product) and on CPU stand-ins in the gloo tests.
    """

    def __init__(self, local_search: Callable, merge: Callable, group=None):
        self.local_search = local_search
        self.merge = merge
        self.group = group

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def recall(self, queries: torch.Tensor, k: int, all_gather_queries: bool = False
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        S = self.world
        nq = queries.shape[0]
        if not dist.is_initialized():
            s, r = self.local_search(queries, k)
            return self.merge(s.unsqueeze(0), r.unsqueeze(0), k)
        if all_gather_queries:
            allq = torch.empty(S * nq, queries.shape[1], dtype=queries.dtype, device=queries.device)
            dist.all_gather_into_tensor(allq, queries.contiguous(), group=self.group)
        else:
            allq = queries
        s, r = self.local_search(allq, k)                       # [n, k] over this shard
        n = allq.shape[0]
        # ONE collective for scores and rows: fp32 scores travel bit-cast to int32 beside the int32
        # rows as a flat [S*2*n, k] tensor (concatenation along dim 0 is the layout both RCCL
        # and gloo accept); nq*k*8 bytes per rank, latency bound
        packed = torch.cat([s.contiguous().view(torch.int32), r.contiguous()], dim=0)   # [2n, k]
        g = torch.empty(S * 2 * n, k, dtype=torch.int32, device=s.device)
        dist.all_gather_into_tensor(g, packed, group=self.group)
        g = g.view(S, 2, n, k)
        gs, gr = g[:, 0].contiguous().view(torch.float32), g[:, 1].contiguous()
        if all_gather_queries:                                   # keep only this rank's queries
            lo = self.rank * nq
            gs = gs[:, lo:lo + nq].contiguous()
            gr = gr[:, lo:lo + nq].contiguous()
        return self.merge(gs, gr, k)


def hip_sharded_recall(hippocampus, row_base: int, now: Optional[float] = None, group=None) -> ShardedRecall:
    """Product wiring: local scan = the HIP kNN over this rank's ``HippocampalFormation`` shard
    (rows reported with ``idx_base = row_base``), merge = ``aura_topk_merge``."""
    from . import ops
    import time as _time

    def local_search(q, k):
        hippocampus._ensure_norms()
        return ops.knn_search(hippocampus.memory_features, hippocampus._inv_norm,
                              hippocampus.memory_metadata, q.contiguous(), k,
                              _time.time() if now is None else now,
                              count=hippocampus.memory_count, idx_base=row_base,
                              check_overflow=True)

    return ShardedRecall(local_search, ops.topk_merge, group)
