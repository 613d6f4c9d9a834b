"""Spiking feed-forward network: Synapsis -> GIF -> Synapsis -> GIF -> mean over T.

Drop-in for ``src/core/language_zone/snn_ffn.py`` (``SNNFFN`` / ``HybridFFN``, same constructor
arguments and ``state_dict`` names: ``syn1, neuron1, syn2, neuron2`` / ``mlp.{0,2}, snn, gate``).

MI355X-first restructuring of ``SNNFFN.forward`` (reference ``snn_ffn.py:55-86``):
  * the reference expands x over T and pushes the T identical copies through GEMM #1
    (``syn1``) and GEMM #2 (``neuron1.linear``): both are computed ONCE per token here and the
    layer-1 GIF kernel reads a time-invariant current (``AURA_GIF_TIME_INVARIANT``) -- exactly
    result-preserving, 4x fewer GEMM FLOPs at T = 4, 16x at T = 16;
  * both GIF time loops are single fused kernels with the state in registers;
  * the layer-2 kernel accumulates the spike mean in registers (``AURA_GIF_MEAN_OUT``), so the
    ``[B*S, T, D]`` spike tensor of layer 2 is never written to HBM.
Dropout is applied as in the reference (identity in eval mode).  When autograd is recording (fp32)
the same restructuring holds -- layer-1 currents are still computed once per token and broadcast
over T (``expand``; autograd sums the T gradients), the loops run through ``GifLoopFunction``
(``aura_gif_train_forward`` / ``aura_gif_backward``) and the mean over T is a torch reduction.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .gif_neuron import (GIFNeuron, _check_input, _need_fp32_training, run_gif_loop,
                         run_gif_loop_grad, wants_grad)
from .synapsis import Synapsis


class SNNFFN(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, output_dim: Optional[int] = None,
                 num_timesteps: int = 4, L: int = 8, dropout: float = 0.1):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim or input_dim
        self.num_timesteps = num_timesteps
        self.syn1 = Synapsis(input_dim, hidden_dim)
        self.neuron1 = GIFNeuron(hidden_dim, hidden_dim, L=L)
        self.syn2 = Synapsis(hidden_dim, self.output_dim)
        self.neuron2 = GIFNeuron(self.output_dim, self.output_dim, L=L)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _check_input(x, "SNNFFN")
        B, S, _ = x.shape
        T = self.num_timesteps
        rows = B * S
        n1, n2 = self.neuron1, self.neuron2
        if wants_grad(self, x):
            _need_fp32_training(x, "SNNFFN")
            h1, _ = self.syn1(x.reshape(rows, 1, self.input_dim), state=None)
            c1 = n1.currents(h1).expand(rows, T, self.hidden_dim)
            spikes1, _ = run_gif_loop_grad(c1, None, decay=n1.decay, L=n1.L, alpha=n1.alpha,
                                           threshold=n1.threshold)
            h2, _ = self.syn2(spikes1, state=None)
            spikes2, _ = run_gif_loop_grad(n2.currents(h2), None, decay=n2.decay, L=n2.L,
                                           alpha=n2.alpha, threshold=n2.threshold)
            return self.dropout(spikes2.mean(dim=1).reshape(B, S, self.output_dim))
        with torch.no_grad():
            # layer 1: currents are identical at every timestep -> one GEMM pair per token
            x1 = x.detach().reshape(rows, 1, self.input_dim)
            h1, _ = self.syn1(x1, state=None)                      # [rows, 1, H]
            c1 = n1.currents(h1).reshape(rows, self.hidden_dim)    # [rows, H]
            spikes1, _ = run_gif_loop(c1, None, decay=n1.decay, L=n1.L, alpha=n1.alpha,
                                      threshold=n1.threshold, T=T, time_invariant=True)
            # layer 2: spikes differ per timestep -> GEMMs over rows*T, mean fused in the loop
            h2, _ = self.syn2(spikes1, state=None)                 # [rows, T, Dout]
            c2 = n2.currents(h2)
            out, _ = run_gif_loop(c2, None, decay=n2.decay, L=n2.L, alpha=n2.alpha,
                                  threshold=n2.threshold, T=T, mean_out=True)
        return self.dropout(out.reshape(B, S, self.output_dim))


class HybridFFN(nn.Module):
    """(1 - sigmoid(gate)) * MLP(x) + sigmoid(gate) * SNNFFN(x)  (reference ``snn_ffn.py:89-145``)."""

    def __init__(self, input_dim: int, hidden_dim: int, snn_ratio: float = 0.5,
                 num_timesteps: int = 4, L: int = 8, dropout: float = 0.1):
        super().__init__()
        self.snn_ratio = snn_ratio
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.GELU(),
                                 nn.Linear(hidden_dim, input_dim), nn.Dropout(dropout))
        self.snn = SNNFFN(input_dim=input_dim, hidden_dim=hidden_dim, output_dim=input_dim,
                          num_timesteps=num_timesteps, L=L, dropout=dropout)
        self.gate = nn.Parameter(torch.tensor(snn_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        mlp_out = self.mlp(x)
        snn_out = self.snn(x)
        g = torch.sigmoid(self.gate)
        return (1 - g) * mlp_out + g * snn_out
