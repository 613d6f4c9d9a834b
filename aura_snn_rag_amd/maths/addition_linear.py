"""AdditionLinear: ``out[b, o] = -||weight_patterns[o] - x[b]||_1 (+ bias[o])``.

API-compatible with the reference's ``src/maths/addition_linear.py`` (parameters
``weight_patterns``, ``learning_signs``, optional ``bias``; uniform init ranges).  The reference
materialises a ``(B, out, in)`` tensor (``addition_linear.py:50-59``); here the projection is one
tiled HIP kernel (``aura_addition_linear``), with ``aura_addition_linear_backward`` behind it when autograd is
recording.  2-D input only, as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class AdditionLinearFunction(torch.autograd.Function):
    """``(x [B, in], weight_patterns [out, in], bias [out] or None) -> [B, out]`` with the gradients autograd derives
    from the reference's ops (``addition_linear.py:50-64``: ``abs`` -> ``sign``, ``sign(0) = 0``)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.addition_linear(x, w, None if bias is None else bias.contiguous())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        g_x, g_w = ops.addition_linear_backward(x, w, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        g_b = g.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return g_x, g_w, g_b


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
        x = input.to(self.weight_patterns.device, torch.float32)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight_patterns.requires_grad or
                                        (self.bias is not None and self.bias.requires_grad)):
            # the reference builds a graph here (input, templates, bias): aura_addition_linear_backward
            return AdditionLinearFunction.apply(x, self.weight_patterns, self.bias)
        b = None if self.bias is None else self.bias.detach().contiguous()
        return ops.addition_linear(x.detach().contiguous(), self.weight_patterns.detach().contiguous(), b)
