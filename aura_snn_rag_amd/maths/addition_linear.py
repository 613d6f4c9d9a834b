"""AdditionLinear: ``out[b, o] = -||weight_patterns[o] - x[b]||_1 (+ bias[o])``.

API-compatible with the reference's ``src/maths/addition_linear.py`` (parameters
``weight_patterns``, ``learning_signs``, optional ``bias``; uniform init ranges).  The reference
materialises a ``(B, out, in)`` tensor (``addition_linear.py:50-59``); here the projection is one
tiled HIP kernel (``aura_addition_linear``).  2-D input only, as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class AdditionLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        shape = (out_features, in_features)
        # randn/ones first, as upstream: a seeded construction then draws the same RNG stream
        self.weight_patterns = nn.Parameter(torch.randn(shape))   # the L1 templates
        self.learning_signs = nn.Parameter(torch.ones(shape))     # kept for state_dict parity (unused in forward)
        self.register_parameter('bias', nn.Parameter(torch.empty(out_features)) if bias else None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        with torch.no_grad():
            self.weight_patterns.uniform_(-0.1, 0.1)
            self.learning_signs.uniform_(-1.0, 1.0)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if input.dim() != 2:
            raise ValueError(f"AdditionLinear expects [batch, {self.in_features}] input, got {tuple(input.shape)}")
        if input.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("AdditionLinear: forward-only HIP path; call under torch.no_grad()")
        x = input.detach().to(self.weight_patterns.device, torch.float32).contiguous()
        b = None if self.bias is None else self.bias.detach().contiguous()
        return ops.addition_linear(x, self.weight_patterns.detach().contiguous(), b)
