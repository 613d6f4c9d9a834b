"""GIF neuron whose input, threshold and adaptation rate follow per-token attention gains.

Drop-in for ``src/core/language_zone/prosody_gif.py``:
``ProsodyModulatedGIF(input_dim, hidden_dim, L, dt, tau, threshold, alpha,
attention_modulation_strength)(x, state=None, attention_gains=None) -> (spikes, (v, theta))``
(argument order of ``prosody_gif.py:33-38``).
The reference's per-timestep Python loop (``prosody_gif.py:64-101``, about a dozen eager ops per
step) is one ``aura_gif_prosody_run`` launch; gains are read once per (row, t).  fp32 and bf16.  When autograd
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
    """Differentiable prosody GIF loop: (h [rows,T,H], gains [rows,T] or None, v0, theta0) -> (spikes, v_T, theta_T).
    One dtype for all four, fp32 or bf16."""

    @staticmethod
    def forward(ctx, h, gains, v0, theta0, decay, L, alpha, threshold, strength):
        h = h.contiguous()
        gains = None if gains is None else gains.detach().contiguous()
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
        dt = save_a.dtype
        g_h = torch.empty_like(save_a)
        g_gains = None if gains is None else torch.zeros(gains.shape, device=gains.device, dtype=torch.float32)
        g_v = g_v.to(dt).contiguous().clone()
        g_theta = g_theta.to(dt).contiguous().clone()
        ops.gif_prosody_backward(save_a, save_th, h, gains, g_spikes.to(dt).contiguous(), g_h, g_gains, g_v, g_theta,
                                 *ctx.cfg)
        if g_gains is not None:
            g_gains = g_gains.to(dt)
        return g_h, g_gains, g_v, g_theta, None, None, None, None, None


class ProsodyModulatedGIF(nn.Module):
    """dtypes follow the reference's eager ops: with the module, the input and the gains all bf16 every op rounds
    to bf16 (``aura_gif_prosody_*_bf16``, bit-identical spikes and state); any fp32 member (fp32 gains with a bf16
    module, a bf16 current from autocast with the fp32 state ``zeros(dtype=x.dtype)``) promotes the loop to fp32
    there, so the bf16 tensors are widened (exactly) and the fp32 kernels run.  One corner differs: a
    caller-supplied bf16 state in such a mixed call is widened before step 0, where the reference still rounds
    ``v * decay`` and ``theta - threshold`` of that first step to bf16."""

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

    def forward(self, x: torch.Tensor, state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                attention_gains: Optional[torch.Tensor] = None):
        _check_input(x, "ProsodyModulatedGIF")
        B, T, _ = x.shape
        H = self.hidden_dim
        if attention_gains is not None and tuple(attention_gains.shape) != (B, T):
            raise ValueError(f"attention_gains: expected {(B, T)}, got {tuple(attention_gains.shape)}")
        if attention_gains is not None and attention_gains.dtype not in (torch.float32, torch.bfloat16):
            attention_gains = attention_gains.to(torch.float32)
        record = wants_grad(self, x, state) or (attention_gains is not None and attention_gains.requires_grad
                                                and torch.is_grad_enabled())
        with torch.enable_grad() if record else torch.no_grad():
            h = self.linear(x)
            if state is None:
                v = torch.zeros(B, H, device=x.device, dtype=x.dtype)
                theta = torch.full((B, H), self.threshold, device=x.device, dtype=x.dtype)
            else:
                v, theta = state
            gains = None if attention_gains is None else attention_gains.to(device=x.device)
            dt = h.dtype
            for t in (v, theta) + (() if gains is None else (gains,)):
                dt = torch.promote_types(dt, t.dtype)
            if dt not in (torch.float32, torch.bfloat16):
                raise TypeError(f"ProsodyModulatedGIF: fp32 and bf16 are implemented, got {dt}")
            h, v, theta = h.to(dt), v.to(dt), theta.to(dt)
            gains = None if gains is None else gains.to(dt)
            cfg = (float(self.decay), int(self.L), float(self.alpha), float(self.threshold),
                   float(self.attention_modulation_strength))
            if record:
                spikes, v, theta = ProsodyGifFunction.apply(h, gains, v, theta, *cfg)
                return spikes, (v, theta)
            h = h.contiguous()
            v, theta = v.contiguous().clone(), theta.contiguous().clone()
            gains = None if gains is None else gains.contiguous()
            spikes = torch.empty_like(h)
            ops.gif_prosody_run(h, gains, spikes, v, theta, *cfg)
        return spikes, (v, theta)
