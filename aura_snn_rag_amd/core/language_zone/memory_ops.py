"""Batched memory retrieval / storage / injection for the RAG transformer layer.

The immediate caller of the hot path in the reference is ``MemoryAugmentedLayer``
(``src/core/language_zone/memory_augmented_layer.py:86-203``): it loops over the batch in Python,
issues one ``retrieve_similar_memories`` per row and then copies ``k`` bank rows one by one.
These functions keep the method signatures and results of that layer and do the work in one
batched recall + one row gather.

Use either as free functions or through ``BatchedMemoryMixin``::

    class FastLayer(BatchedMemoryMixin, MemoryAugmentedLayer):   # reference layer, HIP memory path
        pass
"""
from __future__ import annotations

import uuid
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def retrieve_memories(hippocampus, query: torch.Tensor, k: int = 5, dtype=None
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``query`` [B, D] (already projected) -> (memory_features [B, k, D], memory_scores [B, k]).

    Same contract as the reference loop (``memory_augmented_layer.py:106-130``): slots beyond the
    number of hits stay zero."""
    B, D = query.shape
    dtype = dtype or query.dtype
    dev = query.device
    feats = torch.zeros(B, k, D, device=dev, dtype=dtype)
    scores = torch.zeros(B, k, device=dev, dtype=dtype)
    if hippocampus is None or hippocampus.memory_count == 0:
        return feats, scores
    s, rows = hippocampus.recall_batch(query.detach().float(), k=k)      # [B, k'] (k' <= k)
    kk = s.shape[1]
    valid = rows >= 0
    feats[:, :kk] = hippocampus.gather_features(rows).to(dtype)            # -1 rows gather zeros
    scores[:, :kk] = torch.where(valid, s, torch.zeros_like(s)).to(dtype)
    return feats, scores


def store_memory(hippocampus, hidden_states: torch.Tensor, event_tag: str = "layer") -> None:
    """Mean-pool each batch item and store it (``memory_augmented_layer.py:132-153``)."""
    if hippocampus is None:
        return
    feats = hidden_states.detach().float().mean(dim=1)                      # [B, D]
    ids = [str(uuid.uuid4())[:8] for _ in range(feats.shape[0])]
    hippocampus.create_episodic_memories(ids, feats)


def inject_concat(hidden_states, memory_features, memory_scores):
    """``memory_injection == "concat"`` (``:178-184``)."""
    w = F.softmax(memory_scores, dim=-1).unsqueeze(-1)
    ctx = (memory_features * w).sum(dim=1, keepdim=True).expand(-1, hidden_states.shape[1], -1)
    return hidden_states + 0.1 * ctx


def inject_gate(hidden_states, memory_features, memory_scores, memory_proj, memory_gate):
    """``memory_injection == "gate"`` (``:186-198``): the score-weighted memory context, projected, enters
    through a sigmoid gate computed from [hidden, context]."""
    w = F.softmax(memory_scores, dim=-1).unsqueeze(-1)
    ctx = (memory_features * w).sum(dim=1, keepdim=True).expand(-1, hidden_states.shape[1], -1)
    ctx = memory_proj(ctx)
    gate = memory_gate(torch.cat([hidden_states, ctx], dim=-1))
    return hidden_states + gate * ctx


def inject_cross_attention(hidden_states, memory_features, memory_norm, memory_attention, dropout):
    """``memory_injection == "cross_attention"`` (``:171-179``): the normed hidden states attend to the
    k retrieved rows."""
    attn_out, _ = memory_attention(query=memory_norm(hidden_states), key=memory_features, value=memory_features)
    return hidden_states + dropout(attn_out)


class MemoryInjection(nn.Module):
    """The injection half of ``MemoryAugmentedLayer`` (``memory_augmented_layer.py:47-60,155-203``) as
    a module of its own, with the reference's submodule names (``memory_norm``, ``memory_attention``,
    ``memory_gate``, ``memory_proj``, ``query_proj``) so that a layer checkpoint's keys load with
    ``strict=False``.  ``forward(hidden_states)`` = batched recall + row gather on the HIP bank
    (``retrieve_memories``) followed by the configured injection."""

    def __init__(self, hippocampus, embedding_dim: int, num_heads: int = 8, dropout: float = 0.1,
                 memory_injection: str = "cross_attention", num_retrieved: int = 5):
        super().__init__()
        if memory_injection not in ("cross_attention", "concat", "gate"):
            raise ValueError("memory_injection must be 'cross_attention', 'concat' or 'gate'")
        self.hippocampus = hippocampus
        self.memory_injection = memory_injection
        self.num_retrieved = num_retrieved
        if memory_injection == "cross_attention":
            self.memory_norm = nn.LayerNorm(embedding_dim)
            self.memory_attention = nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=num_heads,
                                                          dropout=dropout, batch_first=True)
        elif memory_injection == "gate":
            self.memory_gate = nn.Sequential(nn.Linear(embedding_dim * 2, embedding_dim), nn.Sigmoid())
            self.memory_proj = nn.Linear(embedding_dim, embedding_dim)
        self.dropout = nn.Dropout(dropout)
        self.query_proj = nn.Linear(embedding_dim, embedding_dim)

    def retrieve_memories(self, hidden_states: torch.Tensor, k: int = 5):
        query = self.query_proj(hidden_states.mean(dim=1))
        return retrieve_memories(self.hippocampus, query, k=k, dtype=hidden_states.dtype)

    def inject_memories(self, hidden_states, memory_features, memory_scores):
        if self.memory_injection == "cross_attention":
            return inject_cross_attention(hidden_states, memory_features, self.memory_norm, self.memory_attention,
                                          self.dropout)
        if self.memory_injection == "concat":
            return inject_concat(hidden_states, memory_features, memory_scores)
        return inject_gate(hidden_states, memory_features, memory_scores, self.memory_proj, self.memory_gate)

    def forward(self, hidden_states: torch.Tensor, use_memory: bool = True) -> torch.Tensor:
        if use_memory and self.hippocampus is not None and self.hippocampus.memory_count > 0:
            mf, ms = self.retrieve_memories(hidden_states, k=self.num_retrieved)
            hidden_states = self.inject_memories(hidden_states, mf, ms)
        return hidden_states


class BatchedMemoryMixin:
    """Overrides ``retrieve_memories`` / ``store_memory`` of the reference layer with the batched
    HIP path; ``inject_memories`` and everything else stay the layer's own."""

    def retrieve_memories(self, hidden_states: torch.Tensor, k: int = 5):
        query = self.query_proj(hidden_states.mean(dim=1))
        return retrieve_memories(self.hippocampus, query, k=k, dtype=hidden_states.dtype)

    def store_memory(self, hidden_states: torch.Tensor):
        store_memory(self.hippocampus, hidden_states, event_tag=f"layer_{id(self)}")
