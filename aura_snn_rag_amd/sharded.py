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

    def __init__(self, local_search: Callable, merge: Callable, group=None, force_collectives: bool = False):
        self.local_search = local_search
        self.merge = merge
        self.group = group
        # run the gather + merge even with one shard (tests: the collectives' packing and RCCL itself are
        # exercised on a one-GPU box)
        self.force_collectives = force_collectives

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
        if not dist.is_initialized() or (S == 1 and not self.force_collectives):
            return self.local_search(queries, k)     # one shard: its sorted top-k is the result
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


class ShardedHippocampus:
    """A row-sharded episodic bank: rank g owns the global rows
    ``[g * R, (g + 1) * R)`` (``R = local.max_memories``) in its own ``HippocampalFormation``; the
    256-row centroid table is replicated.  Every method is COLLECTIVE: all ranks call it with the same
    arguments (the serving layout: writes and queries are visible to every rank; ``recall_batch`` can
    all-gather per-rank query blocks first).

    * ``write``: slots are planned globally exactly as the single bank plans them (append, then the
      reference's slot-0 overwrite or the FIFO ring); the owner of a slot (``slot // R``) stores the
      row.  With the index ready every rank runs the reference's order-dependent nearest-centroid /
      running-mean update (``hippocampal.py:218-230``) over ALL rows of the batch on its replica of the
      table -- 256 x D per row, no communication, bit-identical replicas -- and owners record the ids.
    * ``rebuild_centroids`` (``:345-377``): initial centroids = rows ``perm[:k]`` gathered by
      all-reduce (each row has exactly one owner), local assign, local per-cluster partial sums and
      counts, ONE all-reduce of ``k x D + k`` floats, means, second local assign, counts all-reduced.
    * ``recall_batch``: local recall (the shard's own prefilter shadow / inverted lists, candidate
      mode included) -> global row ids -> all-gather of ``nq * k * 8`` bytes per rank -> merge.

    ``ops`` is injected (the HIP ops in the product, the CPU stand-ins in the gloo tests)."""

    def __init__(self, local, total_rows: int, ops_module=None, group=None, now_fn=None,
                 force_collectives: bool = False, exchange_bounds: bool = True):
        import time as _time
        from . import ops as _ops
        self.local = local
        self.ops = _ops if ops_module is None else ops_module
        self.group = group
        self.R = int(local.max_memories)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_rows = int(total_rows)
        if self.R * self.world < self.total_rows:
            raise ValueError("shards too small for total_rows")
        self.row_base = self.rank * self.R
        self.memory_count = 0                      # global
        self._write_cursor = 0
        self._now = _time.time if now_fn is None else now_fn
        # force_collectives: every collective runs even at world size 1 (a one-GPU box then exercises the
        # packing, the merge and RCCL itself); exchange_bounds: the shards' sampled recall bounds are combined
        # before any shard filters (see recall_batch)
        self.force_collectives = bool(force_collectives) and dist.is_initialized()
        self.exchange_bounds = bool(exchange_bounds)
        self._recall = ShardedRecall(self._local_search, self.ops.topk_merge, group, self.force_collectives)
        self._recall_kw = {}
        self._epoch = 0                            # bumped by every collective mutation (write, bulk_write, rebuild)
        self._exch_epoch = -1                      # epoch for which _exch_ok was agreed
        self._exch_ok = False
        self._check_layout = False                 # bulk_write ran: the next write() verifies the shard fill
        self._scratch = None                       # write(): scratch bank of the replicated centroid update
        self.exchanges = 0                         # bound exchanges performed (tests / diagnostics)

    # ------------------------------------------------------------------ helpers
    def _collective(self) -> bool:
        return self.world > 1 or self.force_collectives

    def _all_reduce(self, t: torch.Tensor, op=None) -> torch.Tensor:
        if self._collective():
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op, group=self.group)
        return t

    def _src0(self) -> int:
        """Global rank of the group's rank 0 (``dist.broadcast`` takes global ranks)."""
        return dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def _local_count(self, global_count: int) -> int:
        return max(0, min(self.R, global_count - self.row_base))

    @property
    def use_centroid_index(self) -> bool:
        return self.local.use_centroid_index

    # ------------------------------------------------------------------ write
    def write(self, memory_ids, features) -> None:
        """Collective batched write: identical to ``HippocampalFormation.create_episodic_memories`` on one
        bank of ``total_rows`` rows, including the rebuild every ``centroids_update_interval`` inserts."""
        import numpy as np
        loc = self.local
        feats = loc._features_to_device(features)
        n = len(memory_ids)
        if feats.shape[0] != n:
            raise ValueError(f"{n} ids but {feats.shape[0]} feature rows")
        M = self.total_rows
        if self._check_layout:
            # bulk_write fills shards independently; the slot-ordered path below needs shard g to hold exactly
            # clamp(count - g R, 0, R) rows.  Checked COLLECTIVELY, so that every rank raises (a rank that
            # raised alone would leave the others hanging in the next collective).
            ok = torch.tensor([1 if loc.memory_count == self._local_count(self.memory_count) else 0],
                              dtype=torch.int64, device=loc.memory_features.device)
            ok = int(self._all_reduce(ok, dist.ReduceOp.MIN).item()) if self._collective() else int(ok.item())
            if not ok:
                raise RuntimeError("ShardedHippocampus.write after bulk_write: the shards are not filled in slot order "
                                   "(shard g must hold clamp(count - g R, 0, R) rows); fill every shard completely or "
                                   "keep using bulk_write")
            self._check_layout = False
        self._epoch += 1
        i = 0
        while i < n:
            run = n - i
            if loc.use_centroid_index:
                interval = max(1, int(loc.centroids_update_interval))
                if self.memory_count < M:
                    run = min(run, interval - (self.memory_count % interval), M - self.memory_count)
                elif self.memory_count % interval == 0 and self.memory_count > loc.centroids_k:
                    run = 1
            ids_r, f_r = memory_ids[i:i + run], feats[i:i + run]
            # global slots, as HippocampalFormation._plan_slots
            count = self.memory_count
            n_app = min(run, max(M - count, 0))
            slots = np.empty(run, dtype=np.int64)
            slots[:n_app] = np.arange(count, count + n_app)
            rest = run - n_app
            if rest:
                if loc._overflow == 'reference':
                    slots[n_app:] = 0
                else:
                    slots[n_app:] = (self._write_cursor + np.arange(rest)) % M
                    self._write_cursor += rest
            now = self._now()
            cids = None
            if loc.use_centroid_index and loc._index_ready:
                # the reference's sequential centroid update over the WHOLE run on this rank's replica: the
                # rows pass through a scratch bank (only their centroid ids and the table's new state matter)
                dev, D = feats.device, feats.shape[1]
                sd = loc.memory_locations.shape[1]
                sc = self._scratch
                if sc is None or sc[0].shape[0] < run or sc[0].device != dev:
                    cap = max(run, min(int(loc.centroids_update_interval), 4096))
                    sc = self._scratch = (torch.empty(cap, D, device=dev), torch.empty(cap, sd, device=dev),
                                          torch.empty(cap, 4, device=dev), torch.empty(cap, device=dev),
                                          torch.arange(cap, dtype=torch.int64, device=dev))
                self.ops.bank_write(sc[0], sc[1], sc[2], sc[3], f_r.contiguous(), sc[4][:run],
                                    loc.current_location.to(device=dev, dtype=torch.float32).contiguous(), now,
                                    centroids=loc.centroids, centroid_counts=loc.centroid_counts,
                                    eff_k=min(loc.centroids_k, loc.centroids.shape[0]), distinct_slots=True)
                cids = sc[2][:run, 2].contiguous()
            own = (slots // self.R) == self.rank
            if own.any():
                idx = np.nonzero(own)[0]
                idx_t = torch.from_numpy(idx).to(feats.device)
                n_app_local = int(own[:n_app].sum())
                loc.write_at([ids_r[j] for j in idx], f_r[idx_t], slots[idx] - self.row_base, n_app_local, now,
                             cids=None if cids is None else cids[idx_t])
            self.memory_count = count + n_app
            i += run
            if (loc.use_centroid_index and self.memory_count % loc.centroids_update_interval == 0
                    and self.memory_count > loc.centroids_k):
                self.rebuild_centroids()

    def bulk_write(self, features, first_index: int = 0, id_prefix: str = "bulk-") -> int:
        """Seeding path: every rank passes ITS OWN rows (already routed: e.g. rank g reads the g-th slice
        of the corpus); rows are appended to the local shard, no centroid update.  Returns the new global
        count; call ``rebuild_centroids`` once at the end.  Shards must be filled evenly by the caller
        (global row ids are ``rank * R + local row``).  ``write`` plans slots globally (slot // R owns the row),
        which only agrees with independently filled shards once every shard before the last non-empty one is
        full; the first ``write`` after a ``bulk_write`` checks that collectively and raises on every rank."""
        self.local.bulk_write(features, id_prefix=id_prefix, first_index=first_index, rebuild=False)
        c = torch.tensor([self.local.memory_count], dtype=torch.int64, device=self.local.memory_features.device)
        self.memory_count = int(self._all_reduce(c).item())
        self._epoch += 1
        self._check_layout = True                  # write() plans slots globally: only valid for slot-ordered shards
        return self.memory_count

    # ------------------------------------------------------------------ rebuild
    def rebuild_centroids(self, perm: Optional[torch.Tensor] = None) -> None:
        loc, ops = self.local, self.ops
        if not loc.use_centroid_index:
            return
        dev = loc.memory_features.device
        n_loc = loc.memory_count
        # global row ids of the local rows; with evenly filled shards (bulk_write) the global count is
        # world * n_loc, with the slot-ordered write path it is self.memory_count
        self._epoch += 1
        counts_all = torch.zeros(self.world, dtype=torch.int64, device=dev)   # int64: exact beyond 2^24 rows
        counts_all[self.rank] = n_loc
        counts_all = self._all_reduce(counts_all).cpu()
        n_glob = int(counts_all.sum().item())
        if n_glob == 0:
            return
        k = min(loc.centroids_k, n_glob)
        if perm is None:
            perm = torch.randperm(n_glob)
            if self._collective():                 # rank 0's draw, as one process would draw it
                p = perm.to(dev)
                dist.broadcast(p, src=self._src0(), group=self.group)
                perm = p.cpu()
        if perm.numel() < k:
            raise ValueError(f"rebuild_centroids: perm has {perm.numel()} entries, {k} initial rows are needed")
        # initial centroids: perm indexes the ACTIVE rows in global order (shard after shard)
        starts = torch.cumsum(counts_all, 0) - counts_all          # first active index of every rank
        pk = perm[:k].long()
        mine = (pk >= starts[self.rank]) & (pk < starts[self.rank] + n_loc)
        init = torch.zeros(loc.centroids.shape, device=dev)
        if bool(mine.any()):
            rows = (pk[mine] - starts[self.rank]).to(device=dev, dtype=torch.int32)
            init[torch.nonzero(mine).flatten().to(dev)] = ops.bank_gather(loc.memory_features, rows)
        cent = self._all_reduce(init)              # each of the k rows has exactly one owner: x + 0 + ... = x
        D = cent.shape[1]
        K = cent.shape[0]                            # rows of the (replicated) centroid table
        sums = torch.zeros(K, D, device=dev)
        cnt = torch.zeros(K, device=dev)
        if n_loc:
            assign = ops.kmeans_assign(loc.memory_features, cent, n_loc, k)
            order, seg_off = ops.group_by_cluster(assign, k)
            ops.kmeans_segment_means(loc.memory_features, order, seg_off, sums, k, sums_only=True)
            cnt[:k] = (seg_off[1:k + 1] - seg_off[:k]).to(torch.float32)
        packed = self._all_reduce(torch.cat([sums, cnt.unsqueeze(1)], dim=1))
        sums, cnt = packed[:, :D], packed[:, D]
        has = (cnt > 0).unsqueeze(1)
        cent = torch.where(has, sums / cnt.clamp(min=1.0).unsqueeze(1), cent)     # empty clusters keep theirs
        cent[k:] = 0
        loc.centroids.copy_(cent)
        cnt2 = torch.zeros(loc.centroids_k, device=dev)
        pending = None
        if n_loc:
            assign = ops.kmeans_assign(loc.memory_features, loc.centroids, n_loc, k)
            pending = ops.kmeans_update(loc.memory_features, assign, loc.centroids, k, counts=cnt2,
                                        meta=loc.memory_metadata, update_means=False)
        loc.centroid_counts = self._all_reduce(cnt2)
        loc._index_ready = True
        loc._invalidate_lists()
        if pending is not None and k == 256 and pending[1].numel() == 257:
            loc._ivf_pending = pending

    # ------------------------------------------------------------------ recall
    def _local_search(self, q, k):
        loc = self.local
        kk = k
        if loc.memory_count == 0:
            loc._last_flag = None                           # (no recall ran: nothing is known about empty queries)
            s = torch.full((q.shape[0], kk), float("-inf"), device=q.device)
            return s, torch.full((q.shape[0], kk), -1, dtype=torch.int32, device=q.device)
        # one rank: the bank's own empty-candidate fallback (no second host read for the merged result)
        s, r = loc.recall_batch(q, k=kk, fallback_empty=self.world == 1, **self._recall_kw)
        if s.shape[1] < kk:                        # a shard with fewer than k rows: pad
            pad = kk - s.shape[1]
            s = torch.cat([s, torch.full((s.shape[0], pad), float("-inf"), device=s.device)], dim=1)
            r = torch.cat([r, torch.full((r.shape[0], pad), -1, dtype=torch.int32, device=r.device)], dim=1)
        if self.world > 1:
            r = torch.where(r >= 0, r + self.row_base, r)
            s = torch.where(r >= 0, s, torch.full_like(s, float("-inf")))
        return s.contiguous(), r.contiguous()

    def _exchange(self, b: torch.Tensor) -> torch.Tensor:
        """bounds [n, 2] = {k-th, ceil(k/S)-th largest sampled lower bound on this shard} -> the combined bound
        [n]: max(max over shards of column 0, min over shards of column 1), one all-reduce(MAX)."""
        v = torch.stack([b[:, 0], -b[:, 1]], dim=1).contiguous()
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
        self.exchanges += 1
        return torch.maximum(v[:, 0], -v[:, 1]).contiguous()

    def _exchange_agreed(self) -> bool:
        """Can EVERY shard run the staged inverted-list recall?  (Agreed by one all-reduce per mutation epoch.)"""
        if self._exch_epoch != self._epoch:
            loc = self.local
            ok = bool(loc._candidate_mode() and loc._shadow_applies() and loc.centroids.shape[0] == 256
                      and loc.memory_features.shape[1] % 8 == 0 and loc.memory_features.shape[1] <= 768
                      and loc.memory_count >= 256)
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=loc.memory_features.device)
            self._exch_ok = bool(int(self._all_reduce(t, dist.ReduceOp.MIN).item()))
            self._exch_epoch = self._epoch
        return self._exch_ok

    # ------------------------------------------------------------------ persistence (sharded)
    def bank_state(self) -> dict:
        """Host-side state of THIS rank's shard plus the global counters (``HippocampalFormation.bank_state``);
        save it per rank beside ``local.state_dict()``."""
        return {"world": self.world, "rank": self.rank, "rows_per_rank": self.R, "total_rows": self.total_rows,
                "memory_count": self.memory_count, "write_cursor": self._write_cursor,
                "local": self.local.bank_state()}

    def load_bank_state(self, state: dict) -> None:
        """Inverse of ``bank_state`` (after ``local.load_state_dict``); the sharding must be the one saved."""
        if (int(state["world"]), int(state["rank"]), int(state["rows_per_rank"])) != (self.world, self.rank, self.R):
            raise ValueError("sharded bank_state: saved for rank %s of %s with %s rows per rank" %
                             (state["rank"], state["world"], state["rows_per_rank"]))
        self.local.load_bank_state(state["local"])
        self.total_rows = int(state["total_rows"])
        self.memory_count = int(state["memory_count"])
        self._write_cursor = int(state["write_cursor"])
        self._epoch += 1
        self._check_layout = False

    def recall_batch(self, queries: torch.Tensor, k: int = 5, now: Optional[float] = None,
                     all_gather_queries: bool = False, use_candidates: Optional[bool] = None,
                     check_overflow: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """Global top-k ``(scores [nq, k], GLOBAL rows [nq, k])`` (``-1`` where fewer than k rows can
        score).  ``all_gather_queries``: every rank brings its own ``nq`` queries and receives their
        results (the serving layout of ``bench.py --gpus N``)."""
        q = self.local._features_to_device(queries)
        self._recall_kw = dict(now=self._now() if now is None else now, use_candidates=use_candidates,
                               check_overflow=check_overflow)
        nq = q.shape[0]
        # candidate mode as every rank sees it (index flags + the GLOBAL row count): the collective decisions
        # below must not depend on what one shard happens to hold
        loc0 = self.local
        cand = bool(use_candidates is not False and getattr(loc0, "use_centroid_index", False)
                    and getattr(loc0, "_index_ready", False) and self.memory_count > loc0.centroids_k)
        # The shards' sampled bounds are combined before any shard filters: a shard's
        # threshold for a query is a lower bound of the query's k-th best score over the WHOLE bank -- the k-th
        # largest sampled lower bound on any one shard is, and so is the minimum over the S shards of their
        # ceil(k / S)-th largest (S disjoint shards x ceil(k / S) distinct rows each).  One all-reduce(MAX) of
        # nq x 2 floats per pass of 8192 queries; every shard then keeps ~1/S of the candidates and survivors
        # it would keep against its own bound.  Whether the staged path is taken is agreed collectively once
        # per mutation epoch (every shard must be able to run it), never per call.
        if cand and self.exchange_bounds and self._collective() and int(k) <= 256 and hasattr(self.ops, "Ivf2Staged") \
                and self._exchange_agreed():
            self._recall_kw["bound_exchange"] = (self._exchange, self.world)
        if all_gather_queries and self._collective():
            # every rank sees every query block: the merged result (and the empty-candidate decision
            # below) is then identical on all ranks, so the fallback stays collective-safe.  The centroid
            # table is replicated, so each query is probed ONCE, by the rank that brings it, and its probes
            # ride in the same all-gather as eight extra columns (int32 bits in fp32 lanes).
            # (decided on state every rank shares -- index flags, the GLOBAL row count, the device type --
            #  so that all ranks gather the same payload shape whatever their own shard holds)
            loc = self.local
            probed = cand and loc.centroids.shape[0] == 256 and q.is_cuda and hasattr(self.ops, "centroid_probe")
            ids = self.ops.centroid_probe(q, loc.centroids, min(8, loc.centroids_k)) if probed else None
            if ids is not None:
                payload = torch.cat([q, ids.view(torch.float32)], dim=1).contiguous()
            else:
                payload = q.contiguous()
            # (all ranks take the same branch: index state and device are properties of the sharded bank)
            allp = torch.empty(self.world * nq, payload.shape[1], dtype=q.dtype, device=q.device)
            dist.all_gather_into_tensor(allp, payload, group=self.group)
            if ids is not None:
                allq = allp[:, :q.shape[1]].contiguous()
                self._recall_kw["probe_ids"] = allp[:, q.shape[1]:].contiguous().view(torch.int32)
            else:
                allq = allp
        else:
            allq = q
        s, r = self._recall.recall(allq, int(k))
        # A query is without candidates on EVERY rank only if this rank's flag says so for some query: the
        # host read of the merged result is skipped otherwise (the decision is the same on all ranks whenever
        # such a query exists, so the fallback's collectives stay matched).
        lf = getattr(self.local, "_last_flag", None)
        may_be_empty = self.world > 1 and (lf is None or bool(lf & getattr(self.ops, "KNN_FLAG_NO_CANDIDATES", 64)))
        if cand and check_overflow and may_be_empty and bool((r[:, 0] < 0).any()):
            # no shard had a candidate for these queries: the reference falls back to the full scan (:269-270)
            sel = torch.nonzero(r[:, 0] < 0).flatten()
            self._recall_kw["use_candidates"] = False
            self._recall_kw.pop("probe_ids", None)
            self._recall_kw.pop("bound_exchange", None)
            s2, r2 = self._recall.recall(allq[sel].contiguous(), int(k))
            s[sel], r[sel] = s2, r2
        if allq is not q:
            lo = self.rank * nq
            s, r = s[lo:lo + nq].contiguous(), r[lo:lo + nq].contiguous()
        return s, r


