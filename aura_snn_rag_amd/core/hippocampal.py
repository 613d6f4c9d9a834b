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

import os
import time
from dataclasses import dataclass, field
from collections.abc import Mapping
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


class _EpisodicView(Mapping):
    """``episodic_memories`` of the reference (``:235-239``: a dict id -> EpisodicMemory) without one
    Python object per write: records are built on access from ``id_to_idx`` and the per-slot write
    times.  Read-only; same keys, ``len`` and iteration order as the reference's dict."""

    def __init__(self, owner: "HippocampalFormation"):
        self._o = owner

    def __getitem__(self, mid: str) -> EpisodicMemory:
        slot = self._o.id_to_idx[mid]
        return EpisodicMemory(memory_id=mid, feature_idx=slot, timestamp=float(self._o._slot_time[slot]))

    def __iter__(self):
        return iter(self._o.id_to_idx)

    def __len__(self) -> int:
        return len(self._o.id_to_idx)

    def __contains__(self, mid) -> bool:
        return mid in self._o.id_to_idx


# AURA_NO_HOST_WORD=1: read the recall's flag with a device-to-host copy instead of polling the completion word (A/B runs)
_NO_HOST_WORD = os.environ.get("AURA_NO_HOST_WORD") is not None
# AURA_EXCHANGE_ONCE=1: a sharded recall combines only the sampled bounds (stage 1), not the candidates' (A/B runs)
_EXCHANGE_TWICE = os.environ.get("AURA_EXCHANGE_ONCE") is None


class _IvfState:
    """Inverted lists of the centroid index in the layout ``aura_knn_search_ivf2`` streams: a
    list-sorted bf16 shadow with ``slack`` free entries behind every list, so that writes are
    appended in place (``aura_ivf2_append``) and the lists are re-packed only when the centroids
    are rebuilt or the slack is used up."""

    def __init__(self, max_rows: int, D: int, slack: int, device):
        self.slack = slack
        self.n_alloc = ops.ivf2_alloc_rows(max_rows, slack)
        self.sorted_bf16 = torch.empty(self.n_alloc, D, dtype=torch.bfloat16, device=device)
        self.sorted_rows = torch.empty(self.n_alloc, dtype=torch.int32, device=device)
        self.pad_off = torch.zeros(257, dtype=torch.int32, device=device)
        self.list_len = torch.zeros(256, dtype=torch.int32, device=device)
        self.pos_of_row = torch.full((max_rows,), -1, dtype=torch.int32, device=device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.n_sorted = 16            # sorted rows in use (host-side bound, a multiple of 16; fixed between re-packs)
        self.appended = 0             # rows appended since the last re-pack
        self.valid = False
        # cached score constants of the sorted rows (aura_ivf2_row_constants): valid for ONE fp32 `now` (the
        # reference's fp32 timestamps give time a 128-second grain) while metadata, rho and layout are unchanged
        self.plan = None              # ops.Ivf2Plan of this layout (validated once, see recall_batch)
        self.rowc: Optional[torch.Tensor] = None
        self.rowc_live = False
        self.rowc_now = 0.0           # the fp32 `now` the table was built for
        self.rowc_version = -1        # memory_metadata._version it was built from (catches in-place edits by the user)


class HippocampalFormation(nn.Module):
    # above this many rows the inverted lists (each probed list read once per batch) beat a masked
    # pass over every row
    MASKED_SCAN_MAX_ROWS = 250_000
    MASKED_SCAN_MAX_QUERIES = 512      # beyond: every 256 queries cost another pass over all rows
    SHADOW_MIN_ROWS = 8192             # below: the all-fp32 scan (no prefilter)

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
        # build-side: bf16 copy of the NORMALISED rows for the exact recall's prefilter (half the bytes
        # to stream; results unchanged) and each row's rounding residual (its part of the prefilter's
        # error bound).  Allocated on the first full-scan recall of >= SHADOW_MIN_ROWS rows.
        self._use_shadow = bool(bf16_shadow) and feature_dim % 8 == 0 and feature_dim <= 768
        self._shadow = None
        self._rho = None
        self._shadow_valid_upto = 0

        self.id_to_idx: Dict[str, int] = {}
        self._idx_to_id: List[Optional[str]] = [None] * max_memories  # dense reverse map
        self._slot_time = np.zeros(max_memories, dtype=np.float64)     # host clock of each slot's last write
        self.episodic_memories = _EpisodicView(self)
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
        self._last_flag = None                    # overflow flag of the last candidate-mode recall that read it
        # inverted lists of the centroid index, derived from memory_metadata[:, 2]:
        self._ivf: Optional[_IvfState] = None     # list-sorted bf16 shadow, kept current by the writes
        self._ivf_pending = None                  # (order, seg_off) left by rebuild_centroids for the next re-pack
        self._lists = None                        # fp32 fallback lists (list_rows, list_off, list_len, longest)
        self._lists_dirty = True

        if overflow not in ('reference', 'fifo'):
            raise ValueError("overflow must be 'reference' or 'fifo'")
        self._overflow = overflow
        self._write_cursor = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate_norms())

    # ------------------------------------------------------------------ plumbing
    def _invalidate_norms(self) -> None:
        self._norms_valid_upto = 0
        self._shadow_valid_upto = 0
        self._invalidate_lists()

    def _invalidate_lists(self) -> None:
        self._lists_dirty = True
        self._ivf_pending = None
        if self._ivf is not None:
            self._ivf.valid = False

    def _ensure_rho(self) -> torch.Tensor:
        dev = self.memory_features.device
        if self._rho is None or self._rho.device != dev:
            self._rho = torch.zeros(self.max_memories, dtype=torch.float32, device=dev)
            self._shadow_valid_upto = 0
            if self._ivf is not None:
                self._ivf.valid = False
        return self._rho

    def _shadow_applies(self) -> bool:
        return self._use_shadow and self.memory_count >= self.SHADOW_MIN_ROWS and self.memory_features.is_cuda

    def _ensure_shadow(self):
        """bf16 shadow of rows [0, memory_count) (+ their residual norms), or None when it does not apply."""
        if not self._shadow_applies():
            return None
        self._ensure_norms()
        rho = self._ensure_rho()
        if self._shadow is None or self._shadow.device != self.memory_features.device:
            self._shadow = torch.empty(self.memory_features.shape, dtype=torch.bfloat16,
                                       device=self.memory_features.device)
            self._shadow_valid_upto = 0
        if self._shadow_valid_upto < self.memory_count:
            lo = self._shadow_valid_upto
            ops.bank_shadow_update(self.memory_features, self._inv_norm, self._shadow, rho, lo,
                                   self.memory_count - lo)
            self._shadow_valid_upto = self.memory_count
        return self._shadow

    def _shadow_after_write(self, slot_t: torch.Tensor, lo: int, hi: int, contiguous: bool) -> None:
        """Keep the shadow current for the rows just written.  The watermark ``_shadow_valid_upto`` only
        ever moves over rows that HAVE been converted: a contiguous append that starts at the watermark
        advances it; rows below it are converted in place; anything else is left to ``_ensure_shadow``
        (which converts [watermark, count) before the next recall)."""
        if self._shadow is None:
            return
        upto = self._shadow_valid_upto
        if contiguous and lo == upto:
            ops.bank_shadow_update(self.memory_features, self._inv_norm, self._shadow, self._rho, lo, hi + 1 - lo)
            self._shadow_valid_upto = hi + 1
        elif lo < upto:
            below = slot_t if hi < upto else slot_t[slot_t < upto]
            ops.bank_shadow_update(self.memory_features, self._inv_norm, self._shadow, self._rho, slots=below.contiguous())

    def _ensure_lists(self):
        """fp32 inverted lists (the fallback of the two-stage lists): row ids of [0, memory_count) sorted
        by centroid id (rows with id < 0 first), list starts and lengths -- a stable sort + bincount on
        the device, redone only after a write / rebuild changed the assignments."""
        n = self.memory_count
        if self._lists is None or self._lists_dirty or self._lists[4] != n:
            order, seg_off = ops.group_by_cluster(self.memory_metadata[:n, 2], 256)
            lens = (seg_off[1:] - seg_off[:-1]).contiguous()
            # slots per query: no query can collect more rows than the 8 longest lists hold
            # (one host read per list rebuild, not per recall)
            longest = int(torch.topk(lens, min(8, lens.numel())).values.sum().item())
            self._lists = (order, seg_off, lens, longest, n)
            self._lists_dirty = False
        return self._lists[:4]

    def _ensure_ivf(self) -> Optional[_IvfState]:
        """The list-sorted bf16 shadow for the two-stage inverted-list recall (re-packed only after a
        centroid rebuild or when the slack behind the lists is used up); None when it does not apply."""
        if not self._shadow_applies():
            return None
        self._ensure_norms()
        rho = self._ensure_rho()
        slack = ops.ivf2_slack(self.centroids_update_interval)
        st = self._ivf
        D = self.memory_features.shape[1]
        if st is None or st.sorted_bf16.device != self.memory_features.device or st.slack != slack:
            st = self._ivf = _IvfState(self.max_memories, D, slack, self.memory_features.device)
        if not st.valid:
            n = self.memory_count
            if self._ivf_pending is not None and self._ivf_pending[0].numel() == n:
                order, seg_off = self._ivf_pending
            else:
                order, seg_off = ops.group_by_cluster(self.memory_metadata[:n, 2], 256)
            self._ivf_pending = None
            ops.ivf2_layout(order, seg_off, slack, st.sorted_rows, st.pad_off, st.list_len)
            st.n_sorted = min(st.n_alloc, ops.ivf2_alloc_rows(n, slack))
            st.pos_of_row.fill_(-1)
            st.flag.zero_()
            st.rowc_live = False
            ops.bank_shadow_sorted(self.memory_features, self._inv_norm, st.sorted_rows, st.sorted_bf16, rho,
                                   st.pos_of_row, st.n_sorted)
            st.appended = 0
            st.valid = True
        return st

    def _ivf_row_constants(self, st: _IvfState, now: float) -> torch.Tensor:
        """The sorted rows' score constants for ``now``, rebuilt only when ``now`` (as fp32), the metadata or
        the list layout changed since the last call (31 us per call at 1 M rows otherwise)."""
        nowf = float(np.float32(now))
        if st.rowc is None or st.rowc.device != st.sorted_rows.device:
            st.rowc = torch.empty(st.n_alloc, 4, dtype=torch.float32, device=st.sorted_rows.device)
            st.rowc_live = False
        ver = self.memory_metadata._version
        if not (st.rowc_live and st.rowc_now == nowf and st.rowc_version == ver):
            ops.ivf2_row_constants(self.memory_metadata, self._rho, st.sorted_rows, st.n_sorted,
                                   self.memory_features.shape[1], nowf, st.rowc)
            st.rowc_live, st.rowc_now, st.rowc_version = True, nowf, ver
        return st.rowc

    def _ivf_after_write(self, uniq_slots: torch.Tensor, n_rows: int, meta_v0: Optional[int] = None) -> None:
        """Append the rows just written to their lists (their centroid ids are in the metadata).
        ``meta_v0``: ``memory_metadata._version`` before the write touched it (the cached score constants
        follow the write only if nobody else edited the metadata since they were built)."""
        st = self._ivf
        if st is None or not st.valid:
            return
        # Rows spread over the 256 lists, so a list's slack lasts far longer than `slack` appended rows;
        # the append kernel flags a list that does overflow (the row is then missing from the lists) and
        # the next recall sees that flag in the read it does anyway (recall_batch).  Holes (overwritten
        # rows) are scanned like padding: re-pack once they are worth it.
        if self._rho is None or st.appended + n_rows > max(st.slack, self.memory_count // 8):
            st.valid = False
            return
        live = st.rowc is not None and st.rowc_live and meta_v0 is not None and st.rowc_version == meta_v0
        ops.ivf2_append(self.memory_features, self._inv_norm, self.memory_metadata, uniq_slots, st.sorted_bf16,
                        st.sorted_rows, st.pad_off, st.list_len, st.pos_of_row, self._rho, st.flag,
                        row_constants=st.rowc if live else None, row_constants_now=st.rowc_now)
        st.appended += n_rows
        st.rowc_live = live
        st.rowc_version = self.memory_metadata._version
        # (n_sorted stays what the re-pack set: pad_off is fixed until the next one, so no list reaches beyond it)

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
        """Recompute the cached row norms and drop the bf16 shadows (call after writing
        ``memory_features`` or ``memory_metadata[:, 2]`` directly)."""
        self._invalidate_norms()
        if self.memory_count:
            self._ensure_norms()

    def _features_to_device(self, features, rows: Optional[int] = None) -> torch.Tensor:
        if (rows is None and isinstance(features, torch.Tensor) and features.dtype == torch.float32 and features.dim() == 2
                and features.device == self.memory_features.device and features.shape[1] == self.memory_features.shape[1]
                and features.is_contiguous() and not features.requires_grad):
            return features                       # already what the kernels take (the common case of batched recall)
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
        """Slots (int64 ndarray) for the next n writes, how many of them append, and the counters they
        leave behind (nothing is mutated until the kernel launch has been accepted)."""
        M, count, cursor = self.max_memories, self.memory_count, self._write_cursor
        n_app = min(n, max(M - count, 0))
        slots = np.empty(n, dtype=np.int64)
        slots[:n_app] = np.arange(count, count + n_app, dtype=np.int64)
        rest = n - n_app
        if rest:
            if self._overflow == 'reference':
                slots[n_app:] = 0                  # count % max_memories with count == max_memories (:200-202)
            else:
                slots[n_app:] = (cursor + np.arange(rest, dtype=np.int64)) % M
                cursor += rest
        return slots, n_app, count + n_app, cursor

    @staticmethod
    def _last_occurrences(slots: np.ndarray, n_app: int) -> Optional[np.ndarray]:
        """Indices (ascending) of the last write to every distinct slot, or None if all are distinct."""
        if slots.size - n_app <= 0 or (slots.size - n_app == 1 and n_app == 0):
            return None
        _, first_rev = np.unique(slots[::-1], return_index=True)
        if first_rev.size == slots.size:
            return None
        return np.sort(slots.size - 1 - first_rev)

    def _after_write(self, slot_t: torch.Tensor, uniq_t: torch.Tensor, slots: np.ndarray, c0: int, n_app: int,
                     meta_v0: Optional[int] = None) -> None:
        """Derived state after a write: the appended run [c0, c0 + n_app) and the overwritten slots."""
        if n_app and self._norms_valid_upto >= c0:     # the kernel refreshed 1/||row|| of the written slots
            self._norms_valid_upto = max(self._norms_valid_upto, c0 + n_app)
        if self._shadow is not None:
            if n_app:
                self._shadow_after_write(slot_t[:n_app], c0, c0 + n_app - 1, contiguous=True)
            if slots.size > n_app:
                over = uniq_t[uniq_t < c0] if n_app else uniq_t
                if over.numel():
                    self._shadow_after_write(over, int(slots[n_app:].min()), int(slots[n_app:].max()), contiguous=False)
        self._lists_dirty = True
        self._ivf_pending = None
        self._ivf_after_write(uniq_t, int(uniq_t.numel()), meta_v0)

    def _write_rows(self, ids: Sequence[str], feats: torch.Tensor, now: float) -> None:
        """Write a run of rows that contains no centroid-rebuild boundary."""
        slots, n_app, new_count, new_cursor = self._plan_slots(len(ids))
        self._store_rows(ids, feats, slots, n_app, new_count, new_cursor, now,
                         online=self.use_centroid_index and self._index_ready)

    def write_at(self, ids: Sequence[str], feats: torch.Tensor, slots: np.ndarray, n_app: int, now: float,
                 cids: Optional[torch.Tensor] = None) -> None:
        """Positioned write, the building block of a row-sharded bank (``sharded.ShardedHippocampus``):
        row i goes to local slot ``slots[i]``; the first ``n_app`` slots extend the bank
        (``memory_count .. memory_count + n_app - 1``), the rest overwrite.  No centroid update happens
        here: ``cids`` (fp32 [n], device) are the centroid ids the caller assigned against the
        replicated centroid table (None: -1, "no list")."""
        feats = self._features_to_device(feats)
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        if feats.shape[0] != len(ids) or slots.size != len(ids):
            raise ValueError("write_at: ids, feats and slots disagree")
        if slots.size == 0:
            return
        if n_app and not np.array_equal(slots[:n_app], np.arange(self.memory_count, self.memory_count + n_app)):
            raise ValueError("write_at: the appended slots must continue the bank")
        if slots.min() < 0 or slots.max() >= self.max_memories or slots[n_app:].max(initial=-1) >= self.memory_count + n_app:
            raise ValueError("write_at: slot out of range")
        self._store_rows(ids, feats, slots, n_app, self.memory_count + n_app, self._write_cursor, now,
                         online=False, cids=cids)

    def _store_rows(self, ids, feats, slots: np.ndarray, n_app: int, new_count: int, new_cursor: int, now: float,
                    online: bool, cids: Optional[torch.Tensor] = None) -> None:
        c0 = self.memory_count
        meta_v0 = self.memory_metadata._version
        slot_t = torch.from_numpy(slots).to(self.device)
        eff_k = min(self.centroids_k, self.centroids.shape[0])
        # A batch that overwrites may name a slot more than once (a full bank in the reference's mode
        # sends every write to slot 0): the last write wins, as in the reference's sequential loop.  The
        # serial centroid kernel walks the rows in order; the parallel kernel gets the survivors only.
        # (appends and a FIFO ring that does not lap itself name every slot once: no need to look)
        ring_distinct = self._overflow != 'reference' and slots.size - n_app <= self.max_memories
        keep = None if (slots.size == n_app or ring_distinct) else self._last_occurrences(slots, n_app)
        keep_t = None if keep is None else torch.from_numpy(keep).to(self.device)
        uniq_t = slot_t if keep is None else slot_t[keep_t].contiguous()
        if keep is not None and not online:
            feats_w, slot_w = feats[keep_t].contiguous(), uniq_t
        else:
            feats_w, slot_w = feats, slot_t
        ops.bank_write(self.memory_features, self.memory_locations, self.memory_metadata,
                       self._inv_norm, feats_w, slot_w,
                       self.current_location.to(device=self.device, dtype=torch.float32).contiguous(),
                       now,
                       centroids=self.centroids if online else None,
                       centroid_counts=self.centroid_counts if online else None,
                       eff_k=eff_k if online else 0, distinct_slots=keep is None)
        if cids is not None:
            c = cids.to(device=self.device, dtype=torch.float32)
            self.memory_metadata[uniq_t, 2] = c if keep is None else c[keep_t]
        self.memory_count, self._write_cursor = new_count, new_cursor
        self._after_write(slot_t, uniq_t, slots, c0, n_app, meta_v0)
        self._slot_time[slots] = time.time()
        slot_list = slots.tolist()
        self.id_to_idx.update(zip(ids, slot_list))
        if n_app:
            self._idx_to_id[c0:c0 + n_app] = list(ids[:n_app])
        if len(slot_list) > n_app:
            over = slot_list[n_app:]
            if ring_distinct and over[-1] - over[0] == len(over) - 1:      # one contiguous run of the ring: a slice
                self._idx_to_id[over[0]:over[-1] + 1] = list(ids[n_app:])
            else:
                for mid, slot in zip(ids[n_app:], over):
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
        meta_v0 = self.memory_metadata._version
        slot_t = torch.arange(s0, s0 + n, dtype=torch.int64, device=self.device)
        ops.bank_write(self.memory_features, self.memory_locations, self.memory_metadata,
                       self._inv_norm, feats[:n].contiguous(), slot_t,
                       self.current_location.to(device=self.device, dtype=torch.float32).contiguous(),
                       time.time())
        self.memory_count = s0 + n
        self._after_write(slot_t, slot_t, np.arange(s0, s0 + n, dtype=np.int64), s0, n, meta_v0)
        self._slot_time[s0:s0 + n] = time.time()
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
                     use_candidates: Optional[bool] = None, check_overflow: bool = True,
                     fallback_empty: bool = True,
                     probe_ids: Optional[torch.Tensor] = None, _retry: int = 0,
                     bound_exchange=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Batched recall: ``(scores [nq, k'], rows [nq, k'])`` with ``k' = min(k, count)``;
        rows are bank row indices (int32), ``-1`` where a query has fewer than ``k'`` candidates.

        ``check_overflow`` (default) reads one small tensor back per call (a host sync): the
        prefilter's overflow flag (candidate lists that did not fit: the call is re-run on the fp32
        path, same results) and, in candidate mode, whether some query was left without candidates
        (it then falls back to the full scan, reference ``:269-270``).  Pass False only inside
        latency-critical loops whose data is known to be well behaved.  ``fallback_empty=False``
        leaves such queries at ``-1`` (a shard of a row-sharded bank: the query may have candidates
        on another shard, so the fallback is the caller's decision after the merge).  ``probe_ids``:
        ``probe(queries)`` computed earlier for these queries against the current centroid table (the
        inverted-list path then skips its own probe; other paths ignore it).  ``bound_exchange``
        (``sharded.ShardedHippocampus``): ``(fn, parts)`` -- the inverted-list recall runs in two stages per pass of
        at most 8192 queries and ``fn(bounds [n, 2]) -> bound [n]`` combines every shard's sampled bounds in
        between (a collective: it is called exactly ``2 ceil(nq / 8192)`` times -- sampled bounds, then the filtered
        candidates' bounds; once per pass with ``AURA_EXCHANGE_ONCE`` -- whatever path this bank takes)."""
        self._last_flag = None                        # set only by a candidate-mode recall that read its flag

        def drain_exchanges(nq_: int) -> None:
            # this bank does not take the staged path: keep the shards' collectives matched with neutral bounds
            if bound_exchange is not None:
                step = ops.Ivf2Staged.MAX_QUERIES
                for lo_ in range(0, nq_, step):
                    for _ in range(2 if _EXCHANGE_TWICE else 1):     # (sampled bounds, then the candidates' bounds)
                        bound_exchange[0](torch.full((min(step, nq_ - lo_), 2), -3.0e38, dtype=torch.float32,
                                                     device=self.device))
        if self.memory_count == 0:
            drain_exchanges(queries.shape[0])
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
            drain_exchanges(q.shape[0])
            shadow = self._ensure_shadow() if q_loc is None else None
            return ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata, q, kk, now,
                                  shadow=shadow, rho=self._rho if shadow is not None else None, **kw)
        nprobe = min(8, self.centroids_k)
        scores = rows = ovf = flag_of = None
        full_index = self.centroids.shape[0] == 256
        masked_ok = (q_loc is None and full_index and self.memory_count <= self.MASKED_SCAN_MAX_ROWS and
                     q.shape[0] <= self.MASKED_SCAN_MAX_QUERIES and bound_exchange is None)
        exchanged = False
        shadow = self._ensure_shadow() if masked_ok else None
        if shadow is not None:
            # up to a few hundred thousand rows the candidate restriction is cheapest as probe masks
            # inside the two-stage scan (one pass over the bf16 shadow; 0.15 vs 0.20 ms at 100k x 768,
            # 256 queries); same rows and score bits as the inverted lists
            scores, rows, ovf = ops.knn_search(self.memory_features, self._inv_norm, self.memory_metadata,
                                               q, kk, now, centroids=self.centroids, nprobe=nprobe,
                                               shadow=shadow, rho=self._rho, count=self.memory_count,
                                               check_overflow=False, return_flag=True)
        elif q_loc is None and full_index and kk <= 256:
            ivf = self._ensure_ivf()
            if ivf is not None and not check_overflow and ivf.appended > ivf.slack:
                ivf.valid = False                     # nobody will read the lists' flag: stay within the proven slack
                ivf = self._ensure_ivf()
            if ivf is not None and bound_exchange is not None:
                scores, rows, ovf = self._recall_ivf2_exchanged(q, kk, now, nprobe, ivf, probe_ids, bound_exchange)
                exchanged = True
                if check_overflow and not _NO_HOST_WORD:
                    # the flag through a host-mapped word behind the last stage (as the plan's path below): polled,
                    # not synchronised for
                    hw = getattr(ivf, "_host_flag", None)
                    if hw is None:
                        hw = ivf._host_flag = ops.HostFlag(q.device)
                    hw.signal(ovf)
                    flag_of = hw.wait
            elif ivf is not None:
                # large banks / large batches: inverted lists on the two-stage scan (every probed list is
                # streamed once per 2048 queries from the list-sorted bf16 shadow); same rows and score bits
                rowc = self._ivf_row_constants(ivf, now)
                plan = ivf.plan
                if plan is None or not plan.matches(self.memory_features, self.memory_metadata, self.centroids,
                                                    ivf.sorted_bf16, ivf.n_sorted, rowc) or plan.nprobe != nprobe:
                    plan = ivf.plan = ops.Ivf2Plan(self.memory_features, self._inv_norm, self.memory_metadata, self.centroids,
                                                   nprobe, ivf.sorted_bf16, self._rho, ivf.sorted_rows, ivf.pad_off,
                                                   ivf.list_len, ivf.n_sorted, ivf.flag, rowc)
                scores, rows, ovf = plan.run(q, kk, now, probe_ids=probe_ids)
                if not _NO_HOST_WORD:
                    flag_of = plan.wait_flag                # (the flag arrives through the plan's completion word)
        if not exchanged:
            drain_exchanges(q.shape[0])
        if check_overflow and scores is not None:
            # ONE host read for both conditions: the library's flag carries the overflow bits of the
            # two-stage lists and the "a query has no candidate at all" bit
            f = flag_of() if flag_of is not None else int(ovf.item())
            if f & ops.KNN_FLAG_LISTS_STALE:          # a write outgrew a list's slack: re-pack, then once more
                self._ivf.valid = False
                if _retry < 2 and bound_exchange is None:   # (an exchanged recall is never repeated: collectives)
                    return self.recall_batch(queries, k=k, locations=locations, now=now, use_candidates=use_candidates,
                                             check_overflow=check_overflow, fallback_empty=fallback_empty,
                                             probe_ids=probe_ids, _retry=_retry + 1)
                f |= 1                                # a fresh re-pack that is stale again: never loop, use the fp32 lists
            if f & ~ops.KNN_FLAG_NO_CANDIDATES:
                scores = rows = None                  # candidate lists too long: the fp32 paths below
            elif not (f & ops.KNN_FLAG_NO_CANDIDATES) or not fallback_empty:
                self._last_flag = f                   # (sharded.ShardedHippocampus: was any query left without candidates?)
                return scores, rows
        if scores is None and q_loc is None and full_index:
            # fp32 inverted lists: every probed list is streamed once per batch
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
        if not check_overflow or not fallback_empty:
            return scores, rows
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

    def _recall_ivf2_exchanged(self, q, kk: int, now: float, nprobe: int, ivf: "_IvfState", probe_ids, bound_exchange):
        """Inverted-list recall in passes of at most 8192 queries, each in two stages with the shards' bounds
        combined in between (``aura_knn_search_ivf2_staged``)."""
        fn, parts = bound_exchange
        k2 = max(1, -(-kk // max(int(parts), 1)))
        rowc = self._ivf_row_constants(ivf, now)
        step = ops.Ivf2Staged.MAX_QUERIES
        nq = q.shape[0]
        # results land in ONE pair of tensors (each pass writes its slice); the constructor's tensor checks run once
        # per bank layout, not once per pass and call: the staged recall of a sharded bank is host-bound otherwise
        # (eight ranks: 2 passes x 5 Python-level steps per call against ~1.1 ms of kernels)
        out_s = torch.empty(nq, kk, dtype=torch.float32, device=q.device)
        out_i = torch.empty(nq, kk, dtype=torch.int32, device=q.device)
        sig = (id(self.memory_features), id(self.memory_metadata), id(self.centroids), id(ivf.sorted_bf16), id(rowc),
               ivf.n_sorted, int(kk), int(nprobe))
        validated = getattr(ivf, "_staged_sig", None) == sig
        flag = None
        for lo in range(0, nq, step):
            hi = min(nq, lo + step)
            st = ops.Ivf2Staged(self.memory_features, self._inv_norm, self.memory_metadata, q[lo:hi], kk, now,
                                self.centroids, nprobe, ivf.sorted_bf16, self._rho, ivf.sorted_rows, ivf.pad_off,
                                ivf.list_len, n_sorted=ivf.n_sorted, lists_flag=ivf.flag,
                                probe_ids=None if probe_ids is None else probe_ids[lo:hi],
                                row_constants=rowc, out=(out_s[lo:hi], out_i[lo:hi]), validated=validated)
            ivf._staged_sig = sig
            validated = True
            bound = fn(st.stage1(k2))
            if _EXCHANGE_TWICE:
                # second exchange, on the FILTERED candidates' bounds: the k-th largest lower bound over the shards'
                # candidates (max over shards of each one's k-th, min over shards of each one's ceil(k / S)-th) is
                # close to the global k-th best score itself, so a shard re-scores ~k / S + gap rows per query
                # instead of k + gap -- the refine is half of a shard's share at 8 shards
                bound2 = fn(st.stage2_bounds(bound, k2))
                _, _, f_ = st.stage3(torch.maximum(bound2, bound))
            else:
                _, _, f_ = st.stage2(bound)
            if hi - lo == nq:
                return out_s, out_i, f_
            flag = f_.clone() if flag is None else flag.bitwise_or_(f_)   # stage 1 of the next pass resets the flag
        return out_s, out_i, flag

    def probe(self, queries: torch.Tensor) -> Optional[torch.Tensor]:
        """The centroid probes of ``queries`` ([nq, 8] int32, the 8 nearest of the 256 centroid rows in
        distance order, reference ``:261-262``) for ``recall_batch(..., probe_ids=...)``, or None when the
        index is not in use.  Ranks of a sharded bank share one centroid table, so a query is probed once,
        by the rank that brings it."""
        if not self._candidate_mode() or self.centroids.shape[0] != 256 or not self.memory_features.is_cuda:
            return None
        return ops.centroid_probe(self._features_to_device(queries), self.centroids, min(8, self.centroids_k))

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
                "slot_time": self._slot_time[:n].tobytes(),            # float64 host clock per slot
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
        self._slot_time[:] = 0.0
        st = state.get("slot_time")
        self._slot_time[:n] = np.frombuffer(st, dtype=np.float64)[:n] if st is not None else time.time()
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
        if self._ivf is not None:
            self._ivf.rowc_live = False                 # strengths changed under the cached score constants

    def decay(self, rate: float = 0.01) -> None:
        self.decay_memories(decay_rate=rate)

    def rebuild_centroids(self, perm: Optional[torch.Tensor] = None) -> None:
        """One Lloyd iteration from a random sample of rows (reference ``:345-377``).

        assign (fp32 matrix cores) -> rows grouped by cluster (device sort) -> means as a segmented
        reduction (the bank is read once) -> second assign -> counts + metadata; the second grouping
        is kept for the next re-pack of the inverted lists."""
        if self.memory_count == 0 or not self.use_centroid_index:
            return
        n = self.memory_count
        k = min(self.centroids_k, n)
        if perm is None:
            # The reference draws randperm(n, device=self.device) (:354).  Up to 2^18 rows the draw comes from
            # torch's CPU generator, so a seeded run reproduces the reference's CPU run (the golden vectors);
            # a full CPU permutation of a 1M-row bank costs ~10 ms per rebuild, so larger banks draw from the
            # device generator, as the reference itself does on a GPU.
            perm = torch.randperm(n) if n <= (1 << 18) else torch.randperm(n, device=self.device)
        init = ops.bank_gather(self.memory_features, perm[:k].to(device=self.device, dtype=torch.int32))
        cent = torch.zeros_like(self.centroids)
        cent[:k] = init
        assign = ops.kmeans_assign(self.memory_features, cent, n, k)
        ops.kmeans_update(self.memory_features, assign, cent, k, update_means=True)
        self.centroids.copy_(cent)          # rows >= k stay zero (ref :366-367)
        assign = ops.kmeans_assign(self.memory_features, self.centroids, n, k)
        counts = torch.zeros(self.centroids_k, device=self.device)
        order, seg_off = ops.kmeans_update(self.memory_features, assign, self.centroids, k, counts=counts,
                                           meta=self.memory_metadata, update_means=False)
        self.centroid_counts = counts
        self._index_ready = True
        self._invalidate_lists()
        if k == 256 and seg_off.numel() == 257:
            self._ivf_pending = (order, seg_off)
