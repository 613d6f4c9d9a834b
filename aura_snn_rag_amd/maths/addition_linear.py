"""AdditionLinear: ``out[b, o] = -||weight_patterns[o] - x[b]||_1 (+ bias[o])``.

Drop-in for ``src/maths/addition_linear.py`` (parameters ``weight_patterns``, ``learning_signs``,
optional ``bias``; same uniform init).  The reference materialises a ``(B, out, in)`` tensor
(``addition_linear.py:50-59``); here the projection is one tiled HIP kernel
(``aura_addition_linear``).  2-D input only, as in the reference (3-D raises there too).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class AdditionLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = False):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight_patterns = nn.Parameter(torch.randn(out_features, in_features))
        self.learning_signs = nn.Parameter(torch.ones(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.weight_patterns, -0.1, 0.1)
        nn.init.uniform_(self.learning_signs, -1.0, 1.0)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if input.dim() != 2:
            raise ValueError(f"AdditionLinear expects [batch, {self.in_features}] input, got {tuple(input.shape)}")
        if input.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("AdditionLinear: forward-only HIP path; call under torch.no_grad()")
        x = input.detach().to(self.weight_patterns.device, torch.float32).contiguous()
        return ops.addition_linear(x, self.weight_patterns.detach().contiguous(),
                                   None if self.bias is None else self.bias.detach().contiguous())
