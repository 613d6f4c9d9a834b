"""GIF neuron whose input, threshold and adaptation rate follow per-token attention gains.

Drop-in for ``src/core/language_zone/prosody_gif.py``:
``ProsodyModulatedGIF(input_dim, hidden_dim, L, dt, tau, threshold, alpha,
attention_modulation_strength)(x, attention_gains=None, state=None) -> (spikes, (v, theta))``.
The reference's per-timestep Python loop (``prosody_gif.py:64-101``, about a dozen eager ops per
step) is one ``aura_gif_prosody_run`` launch; gains are read once per (row, t).  fp32.  When autograd
is recording, the loop runs through ``ProsodyGifFunction`` (``aura_gif_prosody_train_forward`` /
``aura_gif_prosody_backward``): gradients reach the input, the linear layer, the carried state and
the attention gains (through the input gain, the threshold scale and the adaptation rate), with
``MultiBitSurrogate``'s triangular window (``gif_neuron.py:16-22``).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from .gif_neuron import _check_input, wants_grad


class ProsodyGifFunction(torch.autograd.Function):
    """Differentiable prosody GIF loop: (h [rows,T,H], gains [rows,T] or None, v0, theta0) -> (spikes, v_T, theta_T)."""

    @staticmethod
    def forward(ctx, h, gains, v0, theta0, decay, L, alpha, threshold, strength):
        h = h.contiguous()
        gains = None if gains is None else gains.detach().to(torch.float32).contiguous()
        v = v0.detach().contiguous().clone()
        theta = theta0.detach().contiguous().clone()
        spikes, save_a, save_th = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        ops.gif_prosody_train_forward(h, gains, spikes, v, theta, save_a, save_th, float(decay), int(L), float(alpha),
                                      float(threshold), float(strength))
        ctx.save_for_backward(save_a, save_th, h, gains if gains is not None else h.new_empty(0))
        ctx.has_gains = gains is not None
        ctx.cfg = (float(decay), int(L), float(alpha), float(threshold), float(strength))
        return spikes, v, theta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_spikes, g_v, g_theta):
        save_a, save_th, h, gains = ctx.saved_tensors
        gains = gains if ctx.has_gains else None
        g_h = torch.empty_like(save_a)
        g_gains = None if gains is None else torch.zeros_like(gains)
        g_v = g_v.contiguous().clone()
        g_theta = g_theta.contiguous().clone()
        ops.gif_prosody_backward(save_a, save_th, h, gains, g_spikes.contiguous(), g_h, g_gains, g_v, g_theta, *ctx.cfg)
        return g_h, g_gains, g_v, g_theta, None, None, None, None, None


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
        B, T, _ = x.shape
        H = self.hidden_dim
        if attention_gains is not None and tuple(attention_gains.shape) != (B, T):
            raise ValueError(f"attention_gains: expected {(B, T)}, got {tuple(attention_gains.shape)}")
        if wants_grad(self, x, state) or (attention_gains is not None and attention_gains.requires_grad
                                          and torch.is_grad_enabled()):
            h = self.linear(x)
            if state is None:
                v = torch.zeros(B, H, device=x.device, dtype=x.dtype)
                theta = torch.full((B, H), self.threshold, device=x.device, dtype=x.dtype)
            else:
                v, theta = state
            gains = None if attention_gains is None else attention_gains.to(device=x.device, dtype=torch.float32)
            spikes, v, theta = ProsodyGifFunction.apply(h, gains, v, theta, self.decay, self.L, self.alpha,
                                                        self.threshold, self.attention_modulation_strength)
            return spikes, (v, theta)
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
