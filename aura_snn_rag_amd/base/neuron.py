"""Izhikevich / AdEx / LIF neuron modules backed by fused HIP time-loop kernels.

Drop-in for the reference's ``src/base/neuron.py`` on the hot path: same class names, constructor
arguments, buffer names (``a,b,c,d,dt`` / ``params`` / ``beta,threshold,slope``), state attributes
(``v,u`` / ``V,w`` / ``mem``) and input-shape handling.  The Python time loops of the reference
(``neuron.py:186-196``, ``:237-248``) and the per-step eager ops of the LIF (``:135-137``) run as
ONE kernel launch each (``aura_snn_rag_amd/csrc/aura_neuron.hip``).

Izhikevich / AdEx: the spikes are a comparison in the reference and carry no autograd history there either; the
loops accept inputs that require grad and return history-free spikes.  The LIF records through its learnable
surrogate (``LifStepFunction``: ``aura_lif_train_forward`` / ``aura_lif_backward``, fp32 and bf16) exactly when the
reference would build a graph.  Tensors must live on the HIP device; there is no CPU path.
"""
from __future__ import annotations

import csv
import json
from pathlib import Path
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops


def _no_grad_input(x: torch.Tensor, who: str) -> None:
    """Izhikevich / AdEx: the reference's spikes are ``(v >= v_peak).to(dtype)`` (``neuron.py:191``, ``:243``) -- a
    comparison: they carry NO autograd history whatever the input requires, so a loss on them sends nothing back
    through these loops in the reference either.  The HIP loop returns the same (history-free) spikes.  What the
    reference has and this does not: history on the CARRIED STATE ``v / u / w`` (nothing on the path
    backpropagates through it; the state tensors here are plain buffers).  (The LIF path with its learnable
    surrogate records history itself and never comes here with a tensor that requires grad.)"""
    return None


def _as_f32_input(x: torch.Tensor, who: str) -> torch.Tensor:
    if not x.is_cuda:
        raise ops.AuraDeviceError(f"{who}: input is on {x.device}; the neuron loops run only as HIP "
                                  f"kernels (no CPU fallback)")
    if x.dtype != torch.float32:
        raise TypeError(f"{who}: fp32 input expected (the reference state follows the input dtype; "
                        f"only fp32 is implemented for this neuron), got {x.dtype}")
    return x.detach().contiguous()


class LearnableSurrogateGradient(torch.autograd.Function):
    """Heaviside forward / fast-sigmoid backward (``neuron.py:70-108``); kept for API parity and
    for callers that build their own differentiable LIF out of torch ops on the GPU."""

    @staticmethod
    def forward(ctx, input, slope):
        ctx.save_for_backward(input, slope)
        return (input > 0).to(input.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        input, slope = ctx.saved_tensors
        grad_input = grad_output * (slope / ((slope * input).abs() + 1.0) ** 2)
        raw = -grad_output * input.abs() * input.sign() / ((slope * input.abs() + 1.0) ** 2)
        extra = raw.ndim - slope.ndim
        grad_slope = raw.sum(dim=list(range(extra))) if (slope.shape != raw.shape and extra > 0) else raw
        return grad_input, grad_slope


def surrogate_spike(x, slope):
    return LearnableSurrogateGradient.apply(x, slope)


class _ScalarCache:
    """Host copies of small parameter buffers, refreshed when the buffer is modified in place
    (``Tensor._version``) or replaced -- avoids a device sync per forward call."""

    def __init__(self):
        self._key = None
        self._vals = None

    def get(self, tensors):
        key = tuple((id(t), t._version, t.device) for t in tensors)
        if key != self._key:
            self._vals = [t.detach().float().cpu().reshape(-1).tolist() for t in tensors]
            self._key = key
        return self._vals


class IzhikevichNeuron(nn.Module):
    """``IzhikevichNeuron(a,b,c,d,dt)(I)``; state ``v,u`` persists across calls while the number
    of neurons is unchanged (``neuron.py:170-172``)."""

    def __init__(self, a=0.02, b=0.2, c=-65.0, d=6.0, dt=0.2):
        super().__init__()
        self.register_buffer("a", torch.tensor(float(a)))
        self.register_buffer("b", torch.tensor(float(b)))
        self.register_buffer("c", torch.tensor(float(c)))
        self.register_buffer("d", torch.tensor(float(d)))
        self.register_buffer("dt", torch.tensor(float(dt)))
        self.v = None
        self.u = None
        self._cache = _ScalarCache()

    def reset_state(self):
        self.v = None
        self.u = None

    def _scalars(self):
        return [x[0] for x in self._cache.get([self.a, self.b, self.c, self.d, self.dt])]

    def forward_sequence(self, I_seq: torch.Tensor) -> torch.Tensor:
        _no_grad_input(I_seq, "IzhikevichNeuron")
        I = _as_f32_input(I_seq, "IzhikevichNeuron")
        if I.dim() == 3:
            B, T, D = I.shape
            n = B * D
        else:
            if I.dim() == 1:
                I = I.unsqueeze(0)
            T = I.shape[-1]
            I = I.reshape(I.shape[0], T)
            n = I.shape[0]
        if self.v is None or self.v.shape[0] != n:
            self.v = torch.full((n,), -65.0, device=I.device, dtype=I.dtype)
            self.u = self.b.to(I.device) * self.v
        a, b, c, d, dt = self._scalars()
        spikes = torch.empty_like(I)
        if I.dim() == 3:
            ops.izh_run_btd(I, spikes, self.v, self.u, a, b, c, d, dt)
        else:
            ops.izh_run_nt(I, spikes, self.v, self.u, a, b, c, d, dt)
        return spikes

    def forward(self, I: torch.Tensor) -> torch.Tensor:
        return self.forward_sequence(I)


class AdExNeuron(nn.Module):
    """Adaptive exponential integrate-and-fire (``neuron.py:202-251``)."""

    def __init__(self, C=200., g_L=10., E_L=-70., V_T=-50., Delta_T=2., tau_w=120., a=0., b=0.,
                 R=1., V_reset=-65., V_spike=30., dt=0.1):
        super().__init__()
        tau_m = C / max(1e-6, g_L)
        self.register_buffer("params", torch.tensor([tau_m, E_L, V_T, Delta_T, R, tau_w, a, b,
                                                     V_reset, V_spike, dt]))
        self.V = None
        self.w = None
        self._cache = _ScalarCache()

    def reset_state(self):
        self.V = None
        self.w = None

    def forward_sequence(self, I_seq: torch.Tensor) -> torch.Tensor:
        _no_grad_input(I_seq, "AdExNeuron")
        I = _as_f32_input(I_seq, "AdExNeuron")
        if I.dim() == 1:
            I = I.unsqueeze(0)
        n = I.shape[0] * I.shape[2] if I.dim() == 3 else I.shape[0]
        params = self._cache.get([self.params])[0]
        if self.V is None or self.V.shape[0] != n:
            self.V = torch.full((n,), params[1], device=I.device, dtype=I.dtype)
            self.w = torch.zeros_like(self.V)
        spikes = torch.empty_like(I)
        if I.dim() == 3:
            ops.adex_run_btd(I, spikes, self.V, self.w, params)
        else:
            ops.adex_run_nt(I, spikes, self.V, self.w, params)
        return spikes

    def forward(self, I: torch.Tensor) -> torch.Tensor:
        return self.forward_sequence(I)


class LifStepFunction(torch.autograd.Function):
    """One differentiable LIF step: (x, mem, beta, threshold, slope) -> (spk, mem_out); backward is
    ``LearnableSurrogateGradient.backward`` of the reference (``neuron.py:80-108``) fused with the
    membrane recurrence, as ``aura_lif_backward``.  All five tensors share one dtype, fp32 or bf16."""

    @staticmethod
    def forward(ctx, x, mem, beta, threshold, slope):
        shape = x.shape
        size = shape[-1]
        x2 = x.contiguous().view(-1, size)
        m2 = mem.detach().contiguous().view(-1, size)
        spk, mem_out, pre = torch.empty_like(x2), torch.empty_like(x2), torch.empty_like(x2)
        ops.lif_train_forward(x2, m2, beta, threshold, spk, mem_out, pre)
        ctx.save_for_backward(pre, beta, threshold, slope.detach())
        ctx.shape = shape
        return spk.view(shape), mem_out.view(shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_spk, g_mem):
        pre, beta, threshold, slope = ctx.saved_tensors
        size = pre.shape[-1]
        g_x, g_prev = torch.empty_like(pre), torch.empty_like(pre)
        raw = torch.empty(pre.shape, device=pre.device, dtype=torch.float32)
        ops.lif_backward(pre, g_spk.to(pre.dtype).contiguous().view(-1, size),
                         g_mem.to(pre.dtype).contiguous().view(-1, size),
                         beta, threshold, slope.contiguous(), g_x, g_prev, raw)
        return g_x.view(ctx.shape), g_prev.view(ctx.shape), None, None, raw.sum(dim=0).to(slope.dtype)


class VectorizedLIFNeuron(nn.Module):
    """One LIF step per call on ``[..., size]`` input; returns ``(spk, mem)``
    (``neuron.py:115-139``).  ``forward_sequence`` runs a whole ``[B, T, size]`` sequence in one
    launch (what ``EnhancedSpikingNeuron`` does with a Python loop,
    ``snn_brain_zones.py:73-79``).

    dtypes follow the reference's eager ops: a module moved to bf16 fed bf16 input computes in bf16 (each op
    rounded, ``aura_lif_*_bf16``); any fp32/bf16 mix promotes to fp32 there (``beta * mem + input``), so the
    bf16 side is widened (exactly) and the fp32 kernels run."""

    def __init__(self, size: int, beta: float = 0.5, threshold: float = 0.6,
                 init_slope: float = 15.0, event_bus: Optional[object] = None, name: str = None):
        super().__init__()
        self.size = size
        self.name = name or "LIF"
        self._event_bus = event_bus
        self.register_buffer("beta", torch.ones(size) * beta)
        self.register_buffer("threshold", torch.ones(size) * threshold)
        self.slope = nn.Parameter(torch.ones(size) * init_slope)
        self.mem = None

    def reset_mem(self):
        self.mem = None

    def _dtype(self, x: torch.Tensor, mem: Optional[torch.Tensor]) -> torch.dtype:
        """The dtype the reference's eager ops would compute this call in: the promotion of the input, the
        buffers and the carried membrane (``mem``: None = none carried, i.e. zeros of the input's dtype)."""
        who = "VectorizedLIFNeuron"
        if not x.is_cuda:
            raise ops.AuraDeviceError(f"{who}: input is on {x.device}; the neuron loops run only as HIP "
                                      f"kernels (no CPU fallback)")
        if x.shape[-1] != self.size:
            raise ValueError(f"VectorizedLIFNeuron(size={self.size}): last dim is {x.shape[-1]}")
        for t, n in ((x, "input"), (self.beta, "beta"), (self.threshold, "threshold"), (self.slope, "slope")):
            if t.dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"{who}: {n} is {t.dtype}; fp32 and bf16 are implemented")
        dt = x.dtype
        for t in (self.beta, self.threshold) + (() if mem is None else (mem,)):
            dt = torch.promote_types(dt, t.dtype)
        return dt

    def _consts(self, dt: torch.dtype):
        return self.beta.to(dt), self.threshold.to(dt)

    def _wants_grad(self, x: torch.Tensor) -> bool:
        return torch.is_grad_enabled() and (x.requires_grad or self.slope.requires_grad or
                                            (self.mem is not None and self.mem.requires_grad))

    def forward(self, input_: torch.Tensor):
        carried = self.mem if (self.mem is not None and self.mem.shape == input_.shape) else None
        dt = self._dtype(input_, carried)
        beta, thr = self._consts(dt)
        if self._wants_grad(input_):
            x = input_.to(dt)
            mem = torch.zeros_like(x) if carried is None else carried.to(dt)
            # the surrogate's slope keeps its own dtype in the reference (it only enters the backward): widen or
            # narrow it for the kernel through autograd so that its gradient comes back in the parameter's dtype
            spk, self.mem = LifStepFunction.apply(x, mem, beta, thr, self.slope.to(dt))
            return spk, self.mem
        x = input_.detach().to(dt).contiguous()
        if carried is None:
            self.mem = torch.zeros_like(x)
        elif carried.requires_grad or carried.dtype != dt or not carried.is_contiguous():
            self.mem = carried.detach().to(dt).contiguous().clone()
        spk = torch.empty_like(x)
        rows = x.numel() // self.size if self.size else 0
        ops.lif_run(x.view(rows, 1, self.size), spk.view(rows, 1, self.size),
                    self.mem.view(rows, self.size), beta, thr)
        return spk, self.mem

    def forward_sequence(self, x_seq: torch.Tensor) -> torch.Tensor:
        if x_seq.dim() != 3:
            raise ValueError("forward_sequence expects [B, T, size]")
        B, T, _ = x_seq.shape
        carried = self.mem if (self.mem is not None and tuple(self.mem.shape) == (B, self.size)) else None
        dt = self._dtype(x_seq, carried)
        beta, thr = self._consts(dt)
        x = x_seq.detach().to(dt).contiguous()
        if carried is None:
            self.mem = torch.zeros(B, self.size, device=x.device, dtype=dt)
        elif carried.requires_grad or carried.dtype != dt or not carried.is_contiguous():
            self.mem = carried.detach().to(dt).contiguous().clone()
        spikes = torch.empty_like(x)
        ops.lif_run(x, spikes, self.mem, beta, thr)
        return spikes


class AdaptiveLIFNeuron(nn.Module):
    """Legacy scalar adapter (``neuron.py:254-266``)."""

    def __init__(self, beta=0.5, threshold=0.6, init_slope=15.0, event_bus=None, name=None):
        super().__init__()
        self.core = VectorizedLIFNeuron(1, beta, threshold, init_slope, event_bus, name)
        self.slope = self.core.slope
        self.threshold = self.core.threshold
        self.beta = self.core.beta

    def forward(self, x):
        if x.dim() == 0:
            x = x.view(1, 1)
        elif x.dim() == 1:
            x = x.view(-1, 1)
        return self.core(x)

    def reset_mem(self):
        self.core.reset_mem()


# ---- preset loaders (host-side utilities, neuron.py:270-326) -------------------------------------

_RS_FALLBACK = {"a": 0.02, "b": 0.2, "c": -65, "d": 6, "I": 14}


def _resolve(path_str: str) -> Path:
    path = Path(path_str)
    if not path.exists():
        alt = Path(__file__).resolve().parents[2] / path.name
        if alt.exists():
            return alt
    return path


def load_izhikevich_presets(csv_path: str) -> Dict[str, Dict[str, float]]:
    """Presets keyed by lower-cased type name; falls back to regular spiking if the file is absent."""
    presets: Dict[str, Dict[str, float]] = {}
    try:
        with _resolve(csv_path).open("r", encoding="utf-8", errors="ignore") as f:
            for row in csv.DictReader(f):
                name = row.get("type") or row.get("Type") or row.get("name") or ""
                if name:
                    presets[name.lower()] = {"a": float(row.get("a", 0.02)), "b": float(row.get("b", 0.2)),
                                             "c": float(row.get("c", -65)), "d": float(row.get("d", 6)),
                                             "I": float(row.get("I", 10))}
    except Exception:
        pass
    if not presets:
        presets["regular spiking (rs)"] = dict(_RS_FALLBACK)
    return presets


def load_izhikevich_patterns_json(json_path: str) -> Dict[str, Dict[str, float]]:
    try:
        with _resolve(json_path).open("r", encoding="utf-8", errors="ignore") as f:
            return json.load(f)
    except Exception:
        return {"regular spiking (rs)": dict(_RS_FALLBACK, dt=0.2)}


def create_izhikevich_from_pattern(name: str, patterns: Dict[str, Dict[str, float]]) -> IzhikevichNeuron:
    p = patterns.get(name) or next(iter(patterns.values()))
    return IzhikevichNeuron(a=p.get("a", 0.02), b=p.get("b", 0.2), c=p.get("c", -65),
                            d=p.get("d", 6), dt=p.get("dt", 0.2))


def simulate_izhikevich(izh: IzhikevichNeuron, T: int = 100, I: float = 10.0) -> torch.Tensor:
    """Constant-current run; the current is created on the neuron's device (the reference builds a
    CPU tensor, ``neuron.py:324-326``)."""
    return izh(torch.full((T,), float(I), device=izh.a.device))
