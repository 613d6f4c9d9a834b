"""Tensor contract of ``NeuromorphicProcessor.process`` (reference ``src/base/snn_processor.py``).

Only the arithmetic of ``_process_through_zones`` (``snn_processor.py:470-542``) is on the hot
path: run each selected zone, softmax the routing weights, combine ``einsum('z,zbd->bd')``.  The
keyword / MoE routing, statistics and plasticity engine of the reference are host-side control
plane and are OUT OF SCOPE; callers pass the zone weights directly or
plug their own router in via ``router``.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import numpy as np
import torch
import torch.nn as nn


def _softmax64(w: np.ndarray) -> np.ndarray:
    # the reference's numpy softmax (src/maths/softmax.py:4-9) at temp = 1
    w = w - np.max(w)
    e = np.exp(w)
    return e / (e.sum() + 1e-12)


class NeuromorphicProcessor(nn.Module):
    def __init__(self, d_model: int = 512, router: Optional[Callable[[str], Dict[str, float]]] = None):
        super().__init__()
        self.d_model = d_model
        self.zone_processors = nn.ModuleDict()
        self.router = router
        self._current_zone_activities: Dict[str, Any] = {}

    def add_zone(self, name: str, zone: nn.Module) -> None:
        self.zone_processors[name] = zone

    def process(self, input: torch.Tensor, context: Optional[Dict[str, Any]] = None,
                zone_weights: Optional[Dict[str, float]] = None) -> torch.Tensor:
        self._current_zone_activities = {}
        if zone_weights is None:
            text = (context or {}).get('text', "")
            if self.router is not None:
                zone_weights = self.router(text)
            else:  # no router: every zone, equal weight (the reference's weights for n active zones)
                n = max(1, len(self.zone_processors))
                zone_weights = {z: 1.0 / n for z in self.zone_processors}
        return self._process_through_zones(input, zone_weights, context)

    def _process_through_zones(self, input: torch.Tensor, zone_weights: Dict[str, float],
                               context: Optional[Dict[str, Any]] = None) -> torch.Tensor:
        outs, ws = [], []
        for name, weight in zone_weights.items():
            if weight > 0.01 and name in self.zone_processors:
                out, activity = self.zone_processors[name](input, context=context)
                outs.append(out)
                ws.append(weight)
                self._current_zone_activities[name] = activity
        if not outs:
            raise RuntimeError("NeuromorphicProcessor: no zone selected (the reference's dense "
                               "fallback MLP is outside the hot path)")
        w = torch.tensor(_softmax64(np.array(ws, dtype=np.float64)), dtype=outs[0].dtype,
                         device=outs[0].device)
        stacked = torch.stack(outs, dim=0)
        if stacked.dim() == 3:
            return torch.einsum('z,zbd->bd', w, stacked)
        if stacked.dim() == 4:
            return torch.einsum('z,zbtd->btd', w, stacked)
        return stacked.mean(dim=0)
