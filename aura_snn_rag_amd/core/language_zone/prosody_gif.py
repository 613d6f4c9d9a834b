"""GIF neuron whose input, threshold and adaptation rate follow per-token attention gains.

Drop-in for ``src/core/language_zone/prosody_gif.py``:
``ProsodyModulatedGIF(input_dim, hidden_dim, L, dt, tau, threshold, alpha,
attention_modulation_strength)(x, attention_gains=None, state=None) -> (spikes, (v, theta))``.
The reference's per-timestep Python loop (``prosody_gif.py:64-101``, about a dozen eager ops per
step) is one ``aura_gif_prosody_run`` launch; gains are read once per (row, t).  fp32, forward only:
under grad mode with anything requiring grad the call raises instead of silently dropping history.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from .gif_neuron import _check_input, wants_grad


class ProsodyModulatedGIF(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, L: int = 16, dt: float = 1.0,
                 tau: float = 10.0, threshold: float = 1.0, alpha: float = 0.01,
                 attention_modulation_strength: float = 0.3):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.L = L
        self.dt = dt
        self.tau = tau
        self.threshold = threshold
        self.alpha = alpha
        self.attention_modulation_strength = attention_modulation_strength
        self.linear = nn.Linear(input_dim, hidden_dim)
        self.decay = math.exp(-dt / tau)

    def forward(self, x: torch.Tensor, attention_gains: Optional[torch.Tensor] = None,
                state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        _check_input(x, "ProsodyModulatedGIF")
        if x.dtype != torch.float32:
            raise TypeError("ProsodyModulatedGIF: fp32 expected")
        if wants_grad(self, x, state) or (attention_gains is not None and attention_gains.requires_grad
                                          and torch.is_grad_enabled()):
            raise NotImplementedError("ProsodyModulatedGIF: forward-only HIP path; call under "
                                      "torch.no_grad() (GIFNeuron has the surrogate-gradient backward)")
        B, T, _ = x.shape
        H = self.hidden_dim
        with torch.no_grad():
            h = self.linear(x).contiguous()
            if state is None:
                v = torch.zeros(B, H, device=x.device, dtype=x.dtype)
                theta = torch.full((B, H), self.threshold, device=x.device, dtype=x.dtype)
            else:
                v = state[0].to(x.dtype).contiguous().clone()
                theta = state[1].to(x.dtype).contiguous().clone()
            gains = None
            if attention_gains is not None:
                if tuple(attention_gains.shape) != (B, T):
                    raise ValueError(f"attention_gains: expected {(B, T)}, got {tuple(attention_gains.shape)}")
                gains = attention_gains.to(device=x.device, dtype=torch.float32).contiguous()
            spikes = torch.empty_like(h)
            ops.gif_prosody_run(h, gains, spikes, v, theta, float(self.decay), int(self.L),
                                float(self.alpha), float(self.threshold),
                                float(self.attention_modulation_strength))
        return spikes, (v, theta)
