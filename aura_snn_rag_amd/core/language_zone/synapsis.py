"""Synapsis: dense synapse ``currents = W . spikes + b`` over a flattened (B*T, in) batch.

API-compatible with the reference's ``src/core/language_zone/synapsis.py`` (constructor keywords,
``weight`` / ``bias`` parameter names, ``(currents, state)`` return, optional trace state).  The
contraction is a plain library GEMM on the matrix cores ("only the
T-dedup is new" -- that lives in ``SNNFFN``); nothing here is a hand-written kernel.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

Trace = Tuple[torch.Tensor, torch.Tensor]


class Synapsis(nn.Module):
    def __init__(self, in_features, out_features, enable_plasticity=False, stdp_lr=0.001,
                 trace_decay=0.95, target_firing_rate=0.3, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.enable_plasticity, self.stdp_lr = enable_plasticity, stdp_lr
        self.trace_decay, self.target_firing_rate = trace_decay, target_firing_rate
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.register_parameter('bias', nn.Parameter(torch.empty(out_features)) if bias else None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        # spiking-regime init: sparser input spikes (lower firing rate) get larger weights
        nn.init.normal_(self.weight, 0.0, 1.0 / math.sqrt(self.in_features * self.target_firing_rate))
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def _fresh_traces(self, like: torch.Tensor, batch: int) -> Trace:
        def z(n):
            return torch.zeros(batch, n, device=like.device, dtype=like.dtype)
        return z(self.in_features), z(self.out_features)

    def forward(self, spikes: torch.Tensor, state: Optional[Trace] = None):
        batch, steps, _ = spikes.shape
        currents = F.linear(spikes.reshape(batch * steps, self.in_features), self.weight, self.bias)
        currents = currents.reshape(batch, steps, self.out_features)
        if not self.enable_plasticity:
            return currents, None
        pre, post = state if state is not None else self._fresh_traces(spikes, batch)
        return currents, self._update_traces(spikes, currents, (pre, post))

    def _update_traces(self, pre_spikes, post_currents, state: Trace) -> Trace:
        """Exponential moving averages of the time-mean pre / post activity."""
        keep, take = self.trace_decay, 1 - self.trace_decay
        return (keep * state[0] + take * pre_spikes.mean(dim=1),
                keep * state[1] + take * post_currents.mean(dim=1))

    def apply_stdp_update(self, pre_trace, post_trace) -> None:
        """dW = lr * outer(mean post trace, mean pre trace), weights clipped to [-10, 10]."""
        if self.enable_plasticity:
            with torch.no_grad():
                self.weight.add_(self.stdp_lr * torch.outer(post_trace.mean(dim=0), pre_trace.mean(dim=0)))
                self.weight.clamp_(-10.0, 10.0)

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"plasticity={self.enable_plasticity}, bias={self.bias is not None}")
