"""Synapsis: dense synapse ``currents = W . spikes + b`` over a flattened (B*T, in) batch.

Drop-in for ``src/core/language_zone/synapsis.py`` (same constructor, ``weight``/``bias``
parameters, SNN-aware init ``std = 1/sqrt(in * target_firing_rate)``, optional trace state).  The
contraction is a plain library GEMM on the matrix cores ("only the T-dedup
is new" -- that lives in ``SNNFFN``).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Synapsis(nn.Module):
    def __init__(self, in_features, out_features, enable_plasticity=False, stdp_lr=0.001,
                 trace_decay=0.95, target_firing_rate=0.3, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.enable_plasticity = enable_plasticity
        self.stdp_lr = stdp_lr
        self.trace_decay = trace_decay
        self.target_firing_rate = target_firing_rate
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        std = 1.0 / math.sqrt(self.in_features * self.target_firing_rate)
        nn.init.normal_(self.weight, mean=0.0, std=std)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, spikes, state=None):
        B, T, _ = spikes.shape
        if self.enable_plasticity and state is None:
            state = (torch.zeros(B, self.in_features, device=spikes.device, dtype=spikes.dtype),
                     torch.zeros(B, self.out_features, device=spikes.device, dtype=spikes.dtype))
        flat = spikes.reshape(B * T, self.in_features)
        currents = F.linear(flat, self.weight, self.bias).reshape(B, T, self.out_features)
        new_state = self._update_traces(spikes, currents, state) if self.enable_plasticity else None
        return currents, new_state

    def _update_traces(self, pre_spikes, post_currents, state):
        pre_trace, post_trace = state
        pre_trace = self.trace_decay * pre_trace + (1 - self.trace_decay) * pre_spikes.mean(dim=1)
        post_trace = self.trace_decay * post_trace + (1 - self.trace_decay) * post_currents.mean(dim=1)
        return (pre_trace, post_trace)

    def apply_stdp_update(self, pre_trace, post_trace):
        if not self.enable_plasticity:
            return
        dw = self.stdp_lr * torch.outer(post_trace.mean(dim=0), pre_trace.mean(dim=0))
        with torch.no_grad():
            self.weight.data += dw
            self.weight.data.clamp_(-10.0, 10.0)

    def extra_repr(self):
        return (f'in_features={self.in_features}, out_features={self.out_features}, '
                f'plasticity={self.enable_plasticity}, bias={self.bias is not None}')
