"""GIF neuron (multi-bit spikes, adaptive threshold) with a fused HIP membrane loop.

Drop-in for ``src/core/language_zone/gif_neuron.py``: ``GIFNeuron(input_dim, hidden_dim, L, dt,
tau, threshold, alpha)(x, state=None) -> (spikes [B,T,H], (v, theta))``; ``linear`` is an
``nn.Linear`` (``linear.weight/bias`` in the state_dict); ``decay``, ``threshold``, ``alpha`` and
``L`` are plain mutable attributes (the reference's tests set ``neuron.decay = 1.0``).

The T-step Python loop of the reference (``gif_neuron.py:54-69``, eight eager ops per step with
``v, theta`` round-tripping memory each op) is one ``aura_gif_run`` launch with the state in
registers.  The dense ``nn.Linear`` stays a library GEMM on the matrix cores.  fp32 and bf16; in
bf16 each op rounds to bf16 exactly as the reference's bf16 tensors do.

Training: when autograd is recording (grad mode on and the input, a parameter or the state
requires grad) the loop runs as ``aura_gif_train_forward[_bf16]`` and its gradient as
``aura_gif_backward[_bf16]`` -- BPTT through the T steps with the triangular surrogate of
``MultiBitSurrogate`` -- wrapped in ``GifLoopFunction``; the ``nn.Linear`` around it is ordinary
autograd.  In bf16 the recording forward is the per-op-rounded inference loop bit for bit and the backward is the
fp32 chain evaluated on the forward's bf16-rounded intermediates (``csrc/aura_train.hip``).
"""
from __future__ import annotations

import math
from typing import Any, Tuple

import torch
import torch.nn as nn

from ... import ops


class MultiBitSurrogate(torch.autograd.Function):
    """floor/clamp forward with the triangular surrogate backward of ``gif_neuron.py:6-22``;
    kept for callers that assemble a differentiable GIF from torch ops on the GPU."""

    @staticmethod
    def forward(ctx, input, L):
        ctx.save_for_backward(input)
        ctx.L = L
        return torch.clamp(torch.floor(input), 0, L)

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        L = ctx.L
        dist = torch.abs(input - torch.round(input))
        scale = torch.clamp(1.0 - 2.0 * dist, 0.0, 1.0)
        in_range = (input >= 0.0) & (input <= L + 1.0)
        return grad_output * in_range.float() * scale, None


def _check_input(x: torch.Tensor, who: str) -> None:
    if not x.is_cuda:
        raise ops.AuraDeviceError(f"{who}: input is on {x.device}; the GIF loop runs only as a HIP "
                                  f"kernel (no CPU fallback)")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{who}: fp32 or bf16 expected, got {x.dtype}")


def wants_grad(module: nn.Module, x: torch.Tensor, state=None) -> bool:
    """True when the reference would record autograd history for this call."""
    if not torch.is_grad_enabled():
        return False
    if x.requires_grad or any(p.requires_grad for p in module.parameters()):
        return True
    return state is not None and any(s.requires_grad for s in state)


def _need_fp32_training(x: torch.Tensor, who: str) -> None:
    """The GIF loop trains in fp32 and bf16 (``aura_gif_train_forward[_bf16]``); anything else is refused."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError(f"{who}: the surrogate-gradient kernels are fp32 / bf16; run {x.dtype} "
                                  f"under torch.no_grad() or train in fp32")


class GifLoopFunction(torch.autograd.Function):
    """Differentiable GIF time loop: (h [rows,T,H], v0, theta0) -> (spikes, v_T, theta_T)."""

    @staticmethod
    def forward(ctx, h, v0, theta0, decay, L, alpha, threshold):
        h = h.contiguous()
        v = v0.detach().contiguous().clone()
        theta = theta0.detach().contiguous().clone()
        spikes, save_a, save_th = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        ops.gif_train_forward(h, spikes, v, theta, save_a, save_th, float(decay), int(L), float(alpha),
                              float(threshold))
        ctx.save_for_backward(save_a, save_th)
        ctx.cfg = (float(decay), int(L), float(alpha), float(threshold))
        return spikes, v, theta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_spikes, g_v, g_theta):
        save_a, save_th = ctx.saved_tensors
        g_h = torch.empty_like(save_a)
        g_v = g_v.contiguous().clone()
        g_theta = g_theta.contiguous().clone()
        ops.gif_backward(save_a, save_th, g_spikes.contiguous(), g_h, g_v, g_theta, *ctx.cfg)
        return g_h, g_v, g_theta, None, None, None, None


def run_gif_loop_grad(h: torch.Tensor, state, *, decay: float, L: int, alpha: float, threshold: float):
    """Autograd-recording twin of ``run_gif_loop`` (fp32 or bf16, h [rows, T, H])."""
    rows, _, H = h.shape
    if state is None:
        v = torch.zeros(rows, H, device=h.device, dtype=h.dtype)
        theta = torch.full((rows, H), threshold, device=h.device, dtype=h.dtype)
    else:
        v, theta = state
    spikes, v, theta = GifLoopFunction.apply(h, v, theta, decay, L, alpha, threshold)
    return spikes, (v, theta)


def run_gif_loop(h: torch.Tensor, state, *, decay: float, L: int, alpha: float, threshold: float,
                 T: int, time_invariant: bool = False, mean_out: bool = False):
    """Shared driver: ``h`` is [rows, T, H] (or [rows, H] if time_invariant)."""
    rows, H = h.shape[0], h.shape[-1]
    if state is None:
        v = torch.zeros(rows, H, device=h.device, dtype=h.dtype)
        theta = torch.full((rows, H), threshold, device=h.device, dtype=h.dtype)
    else:
        v, theta = state
        v = v.detach().to(h.dtype).contiguous().clone()
        theta = theta.detach().to(h.dtype).contiguous().clone()
    out = torch.empty((rows, H) if mean_out else (rows, T, H), device=h.device, dtype=h.dtype)
    ops.gif_run(h.contiguous(), out, v, theta, float(decay), int(L), float(alpha), float(threshold),
                T, time_invariant=time_invariant, mean_out=mean_out)
    return out, (v, theta)


class GIFNeuron(nn.Module):
    def __init__(self, input_dim, hidden_dim, L=16, dt=1.0, tau=10.0, threshold=1.0, alpha=0.01):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.L = L
        self.linear = nn.Linear(input_dim, hidden_dim)
        self.decay = math.exp(-dt / tau)
        self.threshold = threshold
        self.alpha = alpha

    def reset_state(self):
        pass

    def currents(self, x: torch.Tensor) -> torch.Tensor:
        """The neuron's input GEMM (``gif_neuron.py:51``)."""
        return self.linear(x)

    def forward(self, x: torch.Tensor, state=None) -> Tuple[torch.Tensor, Any]:
        _check_input(x, "GIFNeuron")
        if wants_grad(self, x, state):
            _need_fp32_training(x, "GIFNeuron")
            return run_gif_loop_grad(self.currents(x), state, decay=self.decay, L=self.L,
                                     alpha=self.alpha, threshold=self.threshold)
        with torch.no_grad():
            h = self.currents(x.detach())
            return run_gif_loop(h, state, decay=self.decay, L=self.L, alpha=self.alpha,
                                threshold=self.threshold, T=x.shape[1])


class BalancedGIFNeuron(GIFNeuron):
    """Separate rectified excitatory / inhibitory input pathways (``gif_neuron.py:74-117``)."""

    def __init__(self, input_dim, hidden_dim, L=16, dt=1.0, tau=10.0, threshold=1.0, alpha=0.01,
                 inhibition_ratio: float = 0.2):
        super().__init__(input_dim=input_dim, hidden_dim=hidden_dim, L=L, dt=dt, tau=tau,
                         threshold=threshold, alpha=alpha)
        self.inh_ratio = inhibition_ratio
        self.exc_dim = int(hidden_dim * (1.0 - inhibition_ratio))
        self.inh_dim = hidden_dim - self.exc_dim
        self.linear_exc = nn.Linear(input_dim, self.exc_dim)
        self.linear_inh = nn.Linear(input_dim, self.inh_dim)

    def currents(self, x: torch.Tensor) -> torch.Tensor:
        i_exc = torch.relu(self.linear_exc(x))
        i_inh = -torch.relu(self.linear_inh(x))
        return torch.cat([i_exc, i_inh], dim=-1)
