"""HippocampalFormation: episodic memory bank with HIP write / recall / centroid kernels.

Drop-in for the reference's ``src/core/hippocampal.py`` on the hot path.  Same constructor,
method names, return types, public attributes and ``state_dict`` buffer names
(``memory_features, memory_locations, memory_metadata, centroids, centroid_counts, ...``,
reference ``hippocampal.py:57-117``), so checkpoints round-trip.

What runs where
  * ``create_episodic_memory`` -> ``aura_bank_write`` (row store + 1/||row|| + metadata, and the
    online nearest-centroid running mean when the index is ready, ref ``:211-232``);
  * ``retrieve_similar_memories`` -> ``aura_knn_search_ex`` (fp32-MFMA scan with the reference's
    combined-score epilogue + exact top-k; optional centroid-candidate mask, ref ``:259-307``);
  * ``rebuild_centroids`` -> ``aura_kmeans_assign`` / ``aura_kmeans_update`` (ref ``:345-377``);
  * ``decay_memories`` -> ``aura_bank_decay`` (ref ``:334``);
  * place / grid / time-cell rate codes stay plain torch ops on the device (SURVEY.md 8a row a6:
    not on the throughput path).

Batch entry points that the reference lacks (its callers loop in Python,
``memory_augmented_layer.py:113-121``): ``create_episodic_memories`` and ``recall_batch``.

Reference defects on this path (SURVEY.md 8a "hazards") and what this class does:
  * full bank -> always slot 0 (``:200-202``): REPRODUCED by default (``overflow='reference'``);
    ``overflow='fifo'`` opts into a real ring buffer.
  * centroid candidates: ``topk`` positions are looked up as bank rows (``:307-317``) -> right
    scores, wrong ids: FIXED (rows are reported); ``k`` is clamped to the number of candidates
    instead of raising; ``location=`` works with candidates.
  * fp32 timestamps / fp32 ``age`` (``:215,296``): REPRODUCED (deterministic given the clock).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops


@dataclass
class SpatialLocation:
    coordinates: torch.Tensor
    timestamp: float = field(default_factory=time.time)


@dataclass
class EpisodicMemory:
    memory_id: str
    feature_idx: int
    timestamp: float
    strength: float = 1.0


class HippocampalFormation(nn.Module):
    # above this many rows the inverted lists (each probed list read once per batch) beat a masked
    # pass over every row
    MASKED_SCAN_MAX_ROWS = 250_000
    MASKED_SCAN_MAX_QUERIES = 512      # beyond: every 256 queries cost another pass over all rows

    def __init__(self,
                 spatial_dimensions: int = 2,
                 n_place_cells: int = 2000,
                 n_time_cells: int = 100,
                 n_grid_cells: int = 200,
                 max_memories: int = 100000,
                 feature_dim: int = 768,
                 device: str = 'cuda',
                 use_centroid_index: bool = True,
                 overflow: str = 'reference',
                 bf16_shadow: bool = True):
        super().__init__()
        self.spatial_dims = spatial_dimensions
        self.device = torch.device(device if torch.cuda.is_available() else 'cpu')
        dev = self.device

        # place / grid / time cells (reference :55-82)
        self.register_buffer('place_centers', torch.rand(n_place_cells, spatial_dimensions, device=dev) * 20 - 10)
        self.register_buffer('place_radii', torch.rand(n_place_cells, 1, device=dev) * 1.5 + 0.5)
        self.place_max_rate = 20.0
        spacings = torch.logspace(0, 2, n_grid_cells, base=2.0, device=dev).unsqueeze(1)
        self.register_buffer('grid_spacings', spacings)
        self.register_buffer('grid_orientations', torch.rand(n_grid_cells, 1, device=dev) * (torch.pi / 3))
        self.register_buffer('grid_phases', torch.rand(n_grid_cells, spatial_dimensions, device=dev) * spacings)
        self.grid_max_rate = 25.0
        intervals = torch.logspace(0, 3, n_time_cells, base=10.0, device=dev).unsqueeze(1)
        self.register_buffer('time_intervals', intervals)
        self.register_buffer('time_widths', intervals * 0.3)

        # episodic bank, resident in HBM (reference :84-99)
        self.max_memories = max_memories
        self.memory_count = 0
        self.register_buffer('memory_features', torch.zeros(max_memories, feature_dim, device=dev))
        self.register_buffer('memory_locations', torch.zeros(max_memories, spatial_dimensions, device=dev))
        self.register_buffer('memory_metadata', torch.zeros(max_memories, 4, device=dev))
        # build-side: 1/max(||row||, 1e-12), refreshed by every write (not part of the state_dict)
        self.register_buffer('_inv_norm', torch.zeros(max_memories, device=dev), persistent=False)
        self._norms_valid_upto = 0
        # build-side: bf16 copy of the rows for the exact recall's prefilter (half the bytes to
        # stream; results unchanged).  Allocated on the first full-scan recall of >= 8192 rows.
        self._use_shadow = bool(bf16_shadow) and feature_dim % 8 == 0 and feature_dim <= 768
        self._shadow = None
        self._shadow_valid_upto = 0

        self.episodic_memories: Dict[str, EpisodicMemory] = {}
        self.id_to_idx: Dict[str, int] = {}
        self._idx_to_id: List[Optional[str]] = [None] * max_memories  # dense reverse map
        # bulk-ingested rows get implicit ids "<prefix><n>" resolved on lookup: (slot0, slot1, prefix, n0)
        self._implicit_ids: List[Tuple[int, int, str, int]] = []

        self.current_location = torch.zeros(spatial_dimensions, device=dev)
        self.last_event_time = time.time()
        self.register_buffer('k_const', 4 * torch.pi / torch.sqrt(torch.tensor(3.0, device=dev)))

        self.use_centroid_index = use_centroid_index
        self.centroids_k = 256
        self.centroids_update_interval = 512
        self.register_buffer('centroids', torch.zeros(self.centroids_k, feature_dim, device=dev))
        self.register_buffer('centroid_counts', torch.zeros(self.centroids_k, device=dev))
        self._index_ready = False
        # inverted lists (row ids grouped by centroid id), derived lazily from memory_metadata[:, 2]
        self._lists = None            # (list_rows, list_off, list_len) or None
        self._lists2 = None           # (list-sorted bf16 shadow, sorted row ids, padded list starts)
        self._lists_count = -1        # memory_count the lists were built for

        if overflow not in ('reference', 'fifo'):
            raise ValueError("overflow must be 'reference' or 'fifo'")
        self._overflow = overflow
        self._write_cursor = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate_norms())

    # ------------------------------------------------------------------ plumbing
    def _invalidate_norms(self) -> None:
        self._norms_valid_upto = 0
        self._shadow_valid_upto = 0
        self._lists = None
        self._lists2 = None

    def _ensure_sorted_shadow(self):
        """(sorted bf16 rows, sorted row ids, padded list starts) for the two-stage inverted-list
        recall, rebuilt with the lists; None when the shadow does not apply."""
        if not self._use_shadow or self.memory_count < 8192 or not self.memory_features.is_cuda:
            return None
        list_rows, list_off, list_len, _ = self._ensure_lists()
        if getattr(self, "_lists2", None) is None:
            srows, pad_off = ops.ivf2_layout(list_rows, list_off, list_len)
            self._lists2 = (ops.bank_shadow_sorted(self.memory_features, srows), srows, pad_off)
        return self._lists2

    def _ensure_shadow(self):
        """bf16 shadow of rows [0, memory_count), or None when it does not apply."""
        if not self._use_shadow or self.memory_count < 8192 or not self.memory_features.is_cuda:
            return None
        if self._shadow is None or self._shadow.device != self.memory_features.device:
            self._shadow = torch.empty(self.memory_features.shape, dtype=torch.bfloat16,
                                       device=self.memory_features.device)
            self._shadow_valid_upto = 0
        if self._shadow_valid_upto < self.memory_count:
            lo = self._shadow_valid_upto
            ops.bank_shadow_update(self.memory_features, self._shadow, lo, self.memory_count - lo)
            self._shadow_valid_upto = self.memory_count
        return self._shadow

    def _shadow_after_write(self, slot_t: torch.Tensor, lo: int, hi: int) -> None:
        """Keep the shadow current for rows just written (only the part it already covers)."""
        if self._shadow is None or self._shadow_valid_upto <= lo:
            return                                  # those rows get converted by _ensure_shadow
        ops.bank_shadow_update(self.memory_features, self._shadow, slots=slot_t)
        self._shadow_valid_upto = max(self._shadow_valid_upto, min(hi + 1, self.memory_count))

    def _ensure_lists(self):
        """Inverted lists for the IVF recall: row ids of [0, memory_count) sorted by centroid id
        (rows with id < 0 first), list starts and lengths -- a stable sort + bincount on the
        device, redone only after a write / rebuild changed the assignments."""
        n = self.memory_count
        if self._lists is None or self._lists_count != n:
            cids = self.memory_metadata[:n, 2].to(torch.int32)
            order = torch.sort(cids, stable=True).indices.to(torch.int32)
            valid = cids >= 0
            lens = torch.bincount(cids.clamp(min=0).long(), weights=valid.to(torch.float32),
                                  minlength=256)[:256].to(torch.int32)
            n_neg = (n - valid.sum()).to(torch.int32).reshape(1)
            off = torch.cat([n_neg, n_neg + torch.cumsum(lens, 0).to(torch.int32)]).contiguous()
            # slots per query: no query can collect more rows than the 8 longest lists hold
            # (one host read per list rebuild, not per recall)
            longest = int(torch.topk(lens, min(8, lens.numel())).values.sum().item())
            self._lists = (order.contiguous(), off, lens.contiguous(), longest)
            self._lists_count = n
            self._lists2 = None                       # list-sorted bf16 shadow: built on demand
        return self._lists

    def _apply(self, fn, *a, **k):  # keep self.device / current_location in step with .to()
        out = super()._apply(fn, *a, **k)
        self.device = self.memory_features.device
        self.current_location = fn(self.current_location)
        return out

    def _ensure_norms(self) -> None:
        if self._norms_valid_upto < self.memory_count:
            ops.bank_row_norms(self.memory_features, self._inv_norm, 0, self.memory_count)
            self._norms_valid_upto = self.memory_count

    def refresh_norms(self) -> None:
        """Recompute the cached row norms and drop the bf16 shadow (call after writing
        ``memory_features`` directly)."""
        self._norms_valid_upto = 0
        self._shadow_valid_upto = 0
        if self.memory_count:
            self._ensure_norms()

    def _features_to_device(self, features, rows: Optional[int] = None) -> torch.Tensor:
        if isinstance(features, np.ndarray):
            features = torch.from_numpy(features)
        f = features.detach().to(device=self.device, dtype=torch.float32)
        D = self.memory_features.shape[1]
        f = f.reshape(-1, D) if rows is None else f.reshape(rows, D)
        return f.contiguous()

    # ------------------------------------------------------------------ spatial / temporal context
    def update_spatial_state(self, new_location: torch.Tensor, dt: float = 0.1) -> None:
        if isinstance(new_location, np.ndarray):
            new_location = torch.from_numpy(new_location).to(self.device, dtype=torch.float32)
        self.current_location = new_location if new_location.dim() == 1 else new_location[0]

    def get_spatial_context(self) -> Dict[str, Any]:
        loc = self.current_location.unsqueeze(0)
        dists = torch.norm(loc - self.place_centers, dim=1, keepdim=True)
        sigmas = self.place_radii / 3.0
        place_rates = self.place_max_rate * torch.exp(-(dists ** 2) / (2 * sigmas ** 2))
        place_rates = place_rates * (dists <= self.place_radii).float()
        cos_o, sin_o = torch.cos(self.grid_orientations), torch.sin(self.grid_orientations)
        x, y = loc[0, 0], loc[0, 1]
        rotated = torch.cat([cos_o * x - sin_o * y, sin_o * x + cos_o * y], dim=1)
        shifted = rotated - self.grid_phases
        k = self.k_const / self.grid_spacings
        u1 = k * shifted[:, 0:1]
        u2 = k * (-0.5 * shifted[:, 0:1] + 0.866 * shifted[:, 1:2])
        u3 = k * (-0.5 * shifted[:, 0:1] - 0.866 * shifted[:, 1:2])
        grid_val = (torch.cos(u1) + torch.cos(u2) + torch.cos(u3)) / 3.0 + 0.5
        grid_rates = self.grid_max_rate * torch.relu(grid_val)
        return {"current_location": self.current_location, "place_cells": place_rates.flatten(),
                "grid_cells": grid_rates.flatten(), "n_memories": self.memory_count}

    def get_temporal_context(self) -> Dict[str, Any]:
        elapsed = time.time() - self.last_event_time
        diff = elapsed - self.time_intervals
        time_rates = 15.0 * torch.exp(-(diff ** 2) / (2 * (self.time_widths / 3) ** 2))
        return {"time_cells": time_rates.flatten(), "elapsed": elapsed}

    # ------------------------------------------------------------------ write
    def _plan_slots(self, n: int):
        """Slots for the next n writes plus the counters they leave behind (nothing is mutated
        until the kernel launch has been accepted)."""
        count, cursor, slots = self.memory_count, self._write_cursor, []
        for _ in range(n):
            if count >= self.max_memories:
                if self._overflow == 'reference':
                    slots.append(count % self.max_memories)   # == 0, as the reference (:200-202)
                else:
                    slots.append(cursor % self.max_memories)
                    cursor += 1
            else:
                slots.append(count)
                count += 1
        return slots, count, cursor

    def _write_rows(self, ids: Sequence[str], feats: torch.Tensor, now: float) -> None:
        """Write a run of rows that contains no centroid-rebuild boundary."""
        slots, new_count, new_cursor = self._plan_slots(len(ids))
        slot_t = torch.tensor(slots, dtype=torch.int64, device=self.device)
        online = self.use_centroid_index and self._index_ready
        eff_k = min(self.centroids_k, self.centroids.shape[0])
        ops.bank_write(self.memory_features, self.memory_locations, self.memory_metadata,
                       self._inv_norm, feats, slot_t,
                       self.current_location.to(device=self.device, dtype=torch.float32).contiguous(),
                       now,
                       centroids=self.centroids if online else None,
                       centroid_counts=self.centroid_counts if online else None,
                       eff_k=eff_k if online else 0)
        self.memory_count, self._write_cursor = new_count, new_cursor
        self._lists = None
        lo, hi = min(slots), max(slots)
        if self._norms_valid_upto >= lo:      # the kernel refreshed 1/||row|| of the written slots
            self._norms_valid_upto = max(self._norms_valid_upto, hi + 1)
        self._shadow_after_write(slot_t, lo, hi)
        stamp = time.time()
        for mid, slot in zip(ids, slots):
            self.episodic_memories[mid] = EpisodicMemory(memory_id=mid, feature_idx=slot, timestamp=stamp)
            self.id_to_idx[mid] = slot
            self._idx_to_id[slot] = mid

    def id_of_row(self, row: int) -> Optional[str]:
        """Memory id stored at bank row ``row`` (explicit id, else the implicit bulk id)."""
        mid = self._idx_to_id[row]
        if mid is None:
            for s0, s1, prefix, n0 in reversed(self._implicit_ids):
                if s0 <= row < s1:
                    return f"{prefix}{n0 + row - s0}"
        return mid

    def bulk_write(self, features: torch.Tensor, id_prefix: str = "bulk-", first_index: int = 0,
                   rebuild: bool = True) -> int:
        """Seeding path for very large ingests (BASELINE config 5): rows go to the bank through the
        batched write kernel with NO per-row Python objects (ids are implicit,
        ``f"{id_prefix}{first_index + i}"``, resolved by ``id_of_row``) and NO online centroid
        update; with ``rebuild`` the centroid index is rebuilt once at the end.  This deliberately
        departs from the reference's rebuild-every-512-inserts schedule, which is quadratic in the
        bank size; use ``create_episodic_memories`` for reference-identical semantics.
        Returns the number of rows written (stops at ``max_memories``)."""
        feats = self._features_to_device(features)
        n = min(feats.shape[0], self.max_memories - self.memory_count)
        if n <= 0:
            return 0
        s0 = self.memory_count
        slot_t = torch.arange(s0, s0 + n, dtype=torch.int64, device=self.device)
        ops.bank_write(self.memory_features, self.memory_locations, self.memory_metadata,
                       self._inv_norm, feats[:n].contiguous(), slot_t,
                       self.current_location.to(device=self.device, dtype=torch.float32).contiguous(),
                       time.time())
        self.memory_count = s0 + n
        self._lists = None
        if self._norms_valid_upto >= s0:
            self._norms_valid_upto = s0 + n
        self._shadow_after_write(slot_t, s0, s0 + n - 1)
        self._implicit_ids.append((s0, s0 + n, id_prefix, first_index))
        if rebuild and self.use_centroid_index and self.memory_count > self.centroids_k:
            self.rebuild_centroids()
        return n

    def create_episodic_memory(self, memory_id: str, event_id: str, features: torch.Tensor,
                               associated_experts: List[str] = None) -> None:
        """Store one memory (reference ``:195-243``).  ``event_id`` / ``associated_experts`` are
        accepted and unused, as in the reference."""
        self.create_episodic_memories([memory_id], self._features_to_device(features, rows=1))

    def create_episodic_memories(self, memory_ids: Sequence[str], features: torch.Tensor) -> None:
        """Batched one-shot write: identical to calling ``create_episodic_memory`` once per row,
        including the rebuild every ``centroids_update_interval`` inserts (``:242-243``)."""
        feats = self._features_to_device(features)
        n = len(memory_ids)
        if feats.shape[0] != n:
            raise ValueError(f"{n} ids but {feats.shape[0]} feature rows")
        i = 0
        while i < n:
            # run length until the next insert that triggers a rebuild
            run = n - i
            if self.use_centroid_index:
                interval = max(1, int(self.centroids_update_interval))
                if self.memory_count < self.max_memories:
                    to_boundary = interval - (self.memory_count % interval)
                    run = min(run, to_boundary, self.max_memories - self.memory_count)
                elif self.memory_count % interval == 0 and self.memory_count > self.centroids_k:
                    run = 1   # full bank whose size divides the interval: rebuild after every write
            self._write_rows(memory_ids[i:i + run], feats[i:i + run], time.time())
            i += run
            if (self.use_centroid_index and self.memory_count % self.centroids_update_interval == 0
                    and self.memory_count > self.centroids_k):
                self.rebuild_centroids()

    # ------------------------------------------------------------------ recall
    def _candidate_mode(self) -> bool:
        return bool(self.use_centroid_index and self._index_ready and self.memory_count > self.centroids_k)

    def recall_batch(self, queries: torch.Tensor, k: int = 5,
                     locations: Optional[torch.Tensor] = None, now: Optional[float] = None,
                     use_candidates: Optional[bool] = None, check_overflow: bool = True
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Batched recall: ``(scores [nq, k'], rows [nq, k'])`` with ``k' = min(k, count)``;
        rows are bank row indices (int32), ``-1`` where a query has fewer than ``k'`` candidates."""
        if self.memory_count == 0:
            z = torch.empty(queries.shape[0], 0, device=self.device)
            return z, z.to(torch.int32)
        q = self._features_to_device(queries)
        self._ensure_norms()
        kk = min(int(k), self.memory_count)
        now = time.time() if now is None else now
        q_loc = None
        if locations is not None:
            if isinstance(locations, np.ndarray):
                locations = torch.from_numpy(locations)
            q_loc = locations.to(device=self.device, dtype=torch.float32).reshape(-1, self.spatial_dims)
            if q_loc.shape[0] == 1 and q.shape[0] > 1:
                q_loc = q_loc.expand(q.shape[0], -1)
            q_loc = q_loc.contiguous()
        cand = self._candidate_mode() if use_candidates is None else (use_candidates and self._candidate_mode())
        kw = dict(count=self.memory_count, loc=self.memory_locations if q_loc is not None else None,
                  q_loc=q_loc, check_overflow=check_overflow)
        if not cand:
            shadow = self._ensure_shadow() if q_loc is None else None
            return ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata, q, kk, now,
                                  shadow=shadow, **kw)
        nprobe = min(8, self.centroids_k)
        scores = rows = None
        masked_ok = (q_loc is None and self.memory_count <= self.MASKED_SCAN_MAX_ROWS and
                     q.shape[0] <= self.MASKED_SCAN_MAX_QUERIES)
        shadow = self._ensure_shadow() if masked_ok else None
        lists2 = None
        if shadow is None and q_loc is None and self.centroids.shape[0] == 256 and kk <= 256:
            lists2 = self._ensure_sorted_shadow()
        if lists2 is not None:
            # large banks / large batches: inverted lists on the two-stage scan (every probed list is
            # streamed once per 2048 queries from the list-sorted bf16 shadow; 1.9e6 retrievals/s at
            # 1M x 768 vs 0.83e6 for the fp32 lists); same rows and score bits
            sshadow, srows, pad_off = lists2
            _, _, list_len, _ = self._ensure_lists()
            scores, rows, ovf = ops.knn_search_ivf2(self.memory_features, self._inv_norm, self.memory_metadata,
                                                    q, kk, now, self.centroids, nprobe, sshadow, srows,
                                                    pad_off, list_len)
            if check_overflow and int(ovf.item()) != 0:
                scores = rows = None                  # candidate lists too long: fp32 lists below
        if scores is not None:
            pass
        elif shadow is not None and self.centroids.shape[0] == 256:
            # up to a few hundred thousand rows the candidate restriction is cheapest as probe masks
            # inside the two-stage scan (one pass over the bf16 shadow; 0.17 vs 0.20 ms at 100k x 768,
            # 256 queries); same rows and score bits as the inverted lists
            scores, rows = ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata,
                                          q, kk, now, centroids=self.centroids, nprobe=nprobe,
                                          shadow=shadow, **kw)
        elif q_loc is None and self.centroids.shape[0] == 256:
            # inverted-list form: every probed list is streamed once per batch
            list_rows, list_off, list_len, longest = self._ensure_lists()
            cap = ops.ivf_capacity(longest, kk)
            if cap is not None:               # else: lists too long for the two-level select
                scores, rows, _ = ops.knn_search_ivf(self.memory_features, self._inv_norm,
                                                     self.memory_metadata, q, kk, now, self.memory_count,
                                                     self.centroids, nprobe, list_rows, list_off,
                                                     list_len, cap)
        if scores is None:
            scores, rows = ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata,
                                          q, kk, now, centroids=self.centroids, nprobe=nprobe, **kw)
        # a query whose probed centroids own no rows falls back to the full scan (ref :269-270)
        empty = (rows[:, 0] < 0)
        if bool(empty.any()):
            sel = torch.nonzero(empty).squeeze(-1)
            kw2 = dict(kw)
            if q_loc is not None:
                kw2['q_loc'] = q_loc[sel].contiguous()
            s2, r2 = ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata,
                                    q[sel].contiguous(), kk, now, **kw2)
            scores[sel], rows[sel] = s2, r2
        return scores, rows

    def retrieve_similar_memories(self, query_features: torch.Tensor,
                                  location: Optional[torch.Tensor] = None,
                                  k: int = 5) -> List[Tuple[str, float]]:
        """Top-k ``(memory_id, score)`` for one query (reference ``:245-319``)."""
        if self.memory_count == 0:
            return []
        q = self._features_to_device(query_features, rows=1)
        scores, rows = self.recall_batch(q, k=k, locations=location)
        out = []
        for s, r in zip(scores[0].tolist(), rows[0].tolist()):
            mid = self.id_of_row(r) if r >= 0 else None
            if mid is not None:
                out.append((mid, s))
        return out

    # ------------------------------------------------------------------ persistence
    def bank_state(self) -> Dict[str, Any]:
        """Host-side state the reference forgets to checkpoint (``memory_count``, index flag, id
        maps live outside its ``state_dict``, so a reloaded bank reports 0 memories).  Save this
        beside ``state_dict()``; the tensors themselves stay in the ``state_dict`` unchanged."""
        n = self.memory_count
        return {"memory_count": n, "index_ready": bool(self._index_ready),
                "write_cursor": self._write_cursor, "centroids_k": self.centroids_k,
                "centroids_update_interval": self.centroids_update_interval,
                "ids_by_slot": list(self._idx_to_id[:n]),
                "id_to_idx": dict(self.id_to_idx),
                "implicit_ids": list(self._implicit_ids)}

    def load_bank_state(self, state: Dict[str, Any]) -> None:
        """Inverse of ``bank_state`` (call after ``load_state_dict``)."""
        n = int(state["memory_count"])
        if not (0 <= n <= self.max_memories):
            raise ValueError(f"memory_count {n} does not fit a bank of {self.max_memories}")
        self.memory_count = n
        self._index_ready = bool(state.get("index_ready", False))
        self._write_cursor = int(state.get("write_cursor", 0))
        self.centroids_k = int(state.get("centroids_k", self.centroids_k))
        self.centroids_update_interval = int(state.get("centroids_update_interval",
                                                       self.centroids_update_interval))
        self._idx_to_id = list(state["ids_by_slot"]) + [None] * (self.max_memories - n)
        self.id_to_idx = dict(state.get("id_to_idx") or
                              {mid: i for i, mid in enumerate(state["ids_by_slot"]) if mid is not None})
        self._implicit_ids = [tuple(x) for x in state.get("implicit_ids", [])]
        stamp = time.time()
        self.episodic_memories = {mid: EpisodicMemory(memory_id=mid, feature_idx=i, timestamp=stamp)
                                  for mid, i in self.id_to_idx.items()}
        self._invalidate_norms()

    def gather_features(self, rows: torch.Tensor) -> torch.Tensor:
        """``memory_features[rows]`` for int32 rows of any shape (``-1`` -> zeros): the
        ``[B, k, D]`` gather of ``memory_augmented_layer.py:124-128``."""
        return ops.bank_gather(self.memory_features, rows.to(torch.int32).contiguous())

    # ------------------------------------------------------------------ maintenance
    def decay_memories(self, decay_rate: float = 0.01) -> None:
        if self.memory_count == 0:
            return
        ops.bank_decay(self.memory_metadata, float(decay_rate), self.memory_count)

    def decay(self, rate: float = 0.01) -> None:
        self.decay_memories(decay_rate=rate)

    def rebuild_centroids(self, perm: Optional[torch.Tensor] = None) -> None:
        """One Lloyd iteration from a random sample of rows (reference ``:345-377``).  The
        ``randperm`` is drawn from torch's global CPU generator exactly as the reference does on a
        CPU device, so a seeded run reproduces its sample."""
        if self.memory_count == 0 or not self.use_centroid_index:
            return
        n = self.memory_count
        k = min(self.centroids_k, n)
        if perm is None:
            perm = torch.randperm(n)
        init = ops.bank_gather(self.memory_features, perm[:k].to(device=self.device, dtype=torch.int32))
        cent = torch.zeros_like(self.centroids)
        cent[:k] = init
        assign = ops.kmeans_assign(self.memory_features, cent, n, k)
        ops.kmeans_update(self.memory_features, assign, cent, k, update_means=True)
        self.centroids.copy_(cent)          # rows >= k stay zero (ref :366-367)
        assign = ops.kmeans_assign(self.memory_features, self.centroids, n, k)
        counts = torch.zeros(self.centroids_k, device=self.device)
        ops.kmeans_update(self.memory_features, assign, self.centroids, k, counts=counts,
                          meta=self.memory_metadata, update_means=False)
        self.centroid_counts = counts
        self._index_ready = True
        self._lists = None
