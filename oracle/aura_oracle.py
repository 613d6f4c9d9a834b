"""CPU restatement (PyTorch CPU ops, same op order) of the reference hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Each function cites the reference
``file:line`` (relative to the upstream checkout) whose arithmetic it restates.  The op ORDER of
every floating-point expression is the reference's, because ``floor(v/theta)`` and ``v >= 30``
turn a 1-ulp difference into a whole-spike difference ("Hard parts").

Everything here is functional (explicit state in, state out); the stateful reference modules map
onto it as documented per function.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Spiking-neuron loops
# --------------------------------------------------------------------------------------


def izh_run(I: torch.Tensor, v: torch.Tensor, u: torch.Tensor, a, b, c, d, dt
            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Izhikevich Euler loop over ``I[N, T]``; returns ``(spikes[N, T], v[N], u[N])``.

    Restates ``src/base/neuron.py:181-196`` (``_jit_step_loop``).  ``a..dt`` may be Python floats
    or 0-dim tensors (the reference keeps them as fp32 buffers, ``neuron.py:145-149``).
    """
    a, b, c, d, dt = (torch.as_tensor(float(x), dtype=I.dtype) for x in (a, b, c, d, dt))
    out = []
    for t in range(I.shape[1]):
        i_t = I[:, t]
        dv = 0.04 * v * v + 5.0 * v + 140.0 - u + i_t
        v = v + dt * dv
        du = a * (b * v - u)          # uses the UPDATED v (neuron.py:190)
        u = u + dt * du
        spk = (v >= 30.0).to(v.dtype)
        v = torch.where(spk > 0.0, c, v)
        u = torch.where(spk > 0.0, u + d, u)
        out.append(spk)
    return torch.stack(out, dim=1), v, u


def izh_initial_state(n: int, b, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """Lazy state of ``neuron.py:170-172``: v = -65, u = b * v."""
    v = torch.full((n,), -65.0, dtype=dtype)
    u = torch.as_tensor(float(b), dtype=dtype) * v
    return v, u


def flatten_seq(I_seq: torch.Tensor) -> Tuple[torch.Tensor, Optional[Tuple[int, int, int]]]:
    """Shape handling of ``neuron.py:157-167``: ``[T]``, ``[N,T]`` or ``[B,T,D] -> [B*D,T]``."""
    if I_seq.dim() == 3:
        B, T, D = I_seq.shape
        return I_seq.permute(0, 2, 1).reshape(B * D, T), (B, T, D)
    if I_seq.dim() == 1:
        return I_seq.unsqueeze(0), None
    return I_seq.reshape(I_seq.shape[0], I_seq.shape[-1]), None


def unflatten_spikes(spikes: torch.Tensor, btd: Optional[Tuple[int, int, int]]) -> torch.Tensor:
    """``neuron.py:176-178``."""
    if btd is None:
        return spikes
    B, T, D = btd
    return spikes.view(B, D, T).permute(0, 2, 1)


ADEX_DEFAULTS = dict(C=200., g_L=10., E_L=-70., V_T=-50., Delta_T=2., tau_w=120., a=0., b=0.,
                     R=1., V_reset=-65., V_spike=30., dt=0.1)


def adex_params(**kw) -> torch.Tensor:
    """The 11-entry parameter buffer of ``neuron.py:203-207``."""
    p = dict(ADEX_DEFAULTS)
    p.update(kw)
    tau_m = p["C"] / max(1e-6, p["g_L"])
    return torch.tensor([tau_m, p["E_L"], p["V_T"], p["Delta_T"], p["R"], p["tau_w"], p["a"],
                         p["b"], p["V_reset"], p["V_spike"], p["dt"]])


def adex_run(I: torch.Tensor, V: torch.Tensor, w: torch.Tensor, params: torch.Tensor
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """AdEx Euler loop over ``I[N, T]``.  Restates ``src/base/neuron.py:233-248``."""
    tau_m, E_L, V_T, Delta_T, R, tau_w, a, b, V_reset, V_spike, dt = (params[i] for i in range(11))
    out = []
    for t in range(I.shape[1]):
        i_t = I[:, t]
        exp_term = Delta_T * torch.exp((V - V_T) / Delta_T)
        dV = (-(V - E_L) + exp_term - R * w + R * i_t) / tau_m
        V = V + dt * dV
        dw = (a * (V - E_L) - w) / tau_w
        w = w + dt * dw
        spk = (V >= V_spike).to(V.dtype)
        V = torch.where(spk > 0.0, V_reset, V)
        w = torch.where(spk > 0.0, w + b, w)
        out.append(spk)
    return torch.stack(out, dim=1), V, w


def lif_step(x: torch.Tensor, mem: torch.Tensor, beta: torch.Tensor, threshold: torch.Tensor
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One vectorised LIF step.  Restates ``src/base/neuron.py:135-139`` (forward value of the
    surrogate, ``neuron.py:75-77``, is ``(input > 0)``)."""
    mem = beta * mem + x
    pre = mem - threshold
    spk = (pre > 0).to(pre.dtype)                     # the surrogate casts to ITS input's dtype (neuron.py:77)
    mem = mem - spk * threshold
    return spk, mem


def lif_run(x: torch.Tensor, mem: torch.Tensor, beta: torch.Tensor, threshold: torch.Tensor
            ) -> Tuple[torch.Tensor, torch.Tensor]:
    """LIF over a ``[B, T, size]`` sequence, as ``src/base/snn_brain_zones.py:73-79`` drives it."""
    out = []
    for t in range(x.shape[1]):
        s, mem = lif_step(x[:, t], mem, beta, threshold)
        out.append(s)
    return torch.stack(out, dim=1), mem


def gif_run(h: torch.Tensor, v: torch.Tensor, theta: torch.Tensor, decay: float, L: int,
            alpha: float, threshold: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """GIF membrane loop over currents ``h[rows, T, H]`` (after the neuron's own ``nn.Linear``).

    Restates ``src/core/language_zone/gif_neuron.py:54-69`` with the forward value of
    ``MultiBitSurrogate`` (``gif_neuron.py:11-13``).  State dtype follows ``h`` (``:46-47``): in
    bf16 every op rounds to bf16.
    """
    out = []
    for t in range(h.shape[1]):
        i_t = h[:, t, :]
        v = v * decay + i_t
        clamp_limit = L * theta * 2.0
        v = torch.clamp(v, -clamp_limit, clamp_limit)
        normalized_v = v / (theta + 1e-6)
        spike = torch.clamp(torch.floor(normalized_v), 0, L)
        v = v - spike * theta
        if alpha > 0:
            theta = theta + alpha * spike - alpha * (theta - threshold)
        out.append(spike)
    return torch.stack(out, dim=1), v, theta


def gif_initial_state(rows: int, hidden: int, threshold: float, dtype=torch.float32):
    """``gif_neuron.py:45-47``."""
    return (torch.zeros(rows, hidden, dtype=dtype),
            torch.full((rows, hidden), threshold, dtype=dtype))


def gif_decay(dt: float = 1.0, tau: float = 10.0) -> float:
    """``gif_neuron.py:32``."""
    return math.exp(-dt / tau)


def gif_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], *, L: int,
                decay: float, threshold: float = 1.0, alpha: float = 0.01, state=None):
    """Whole ``GIFNeuron.forward`` (``gif_neuron.py:39-71``): linear then loop."""
    rows = x.shape[0]
    hidden = weight.shape[0]
    if state is None:
        v, theta = gif_initial_state(rows, hidden, threshold, x.dtype)
    else:
        v, theta = state
    h = F.linear(x, weight, bias)
    spikes, v, theta = gif_run(h, v, theta, decay, L, alpha, threshold)
    return spikes, (v, theta)


def balanced_gif_forward(x, w_exc, b_exc, w_inh, b_inh, *, L, decay, threshold=1.0, alpha=0.01,
                         state=None):
    """``BalancedGIFNeuron.forward`` (``gif_neuron.py:87-117``): rectified E/I currents per step."""
    rows = x.shape[0]
    hidden = w_exc.shape[0] + w_inh.shape[0]
    if state is None:
        v, theta = gif_initial_state(rows, hidden, threshold, x.dtype)
    else:
        v, theta = state
    hs = []
    for t in range(x.shape[1]):
        i_exc = torch.relu(F.linear(x[:, t, :], w_exc, b_exc))
        i_inh = -torch.relu(F.linear(x[:, t, :], w_inh, b_inh))
        hs.append(torch.cat([i_exc, i_inh], dim=-1))
    h = torch.stack(hs, dim=1)
    spikes, v, theta = gif_run(h, v, theta, decay, L, alpha, threshold)
    return spikes, (v, theta)


def synapsis_forward(spikes: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """``Synapsis.forward`` without plasticity (``synapsis.py:108-120``)."""
    B, T, _ = spikes.shape
    flat = spikes.reshape(B * T, weight.shape[1])
    return F.linear(flat, weight, bias).reshape(B, T, weight.shape[0])


def snnffn_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], *, T: int, L: int,
                   decay: Optional[float] = None, dedup: bool = False) -> torch.Tensor:
    """``SNNFFN.forward`` in eval mode (dropout = identity), ``snn_ffn.py:55-86``.

    ``p`` holds the module's ``state_dict`` entries (``syn1.weight`` ... ``neuron2.linear.bias``).
    ``dedup=True`` computes GEMM #1/#2 once per token instead of T times; the reference's input
    is a T-fold ``expand`` (``snn_ffn.py:69-70``) so the result is bit-identical (checked in
    ``tests/test_oracle_vs_reference.py``).
    """
    decay = gif_decay() if decay is None else decay
    B, S, D = x.shape
    rows = B * S
    if dedup:
        x1 = x.reshape(rows, 1, D)
        h1 = synapsis_forward(x1, p["syn1.weight"], p["syn1.bias"])
        c1 = F.linear(h1, p["neuron1.linear.weight"], p["neuron1.linear.bias"])
        c1 = c1.expand(rows, T, c1.shape[-1])
        v, th = gif_initial_state(rows, c1.shape[-1], 1.0, x.dtype)
        spikes1, _, _ = gif_run(c1, v, th, decay, L, 0.01, 1.0)
    else:
        x_flat = x.unsqueeze(2).expand(-1, -1, T, -1).reshape(rows, T, D)
        h1 = synapsis_forward(x_flat, p["syn1.weight"], p["syn1.bias"])
        spikes1, _ = gif_forward(h1, p["neuron1.linear.weight"], p["neuron1.linear.bias"], L=L,
                                 decay=decay)
    h2 = synapsis_forward(spikes1, p["syn2.weight"], p["syn2.bias"])
    spikes2, _ = gif_forward(h2, p["neuron2.linear.weight"], p["neuron2.linear.bias"], L=L,
                             decay=decay)
    return spikes2.mean(dim=1).reshape(B, S, -1)


def hybridffn_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], *, T: int, L: int) -> torch.Tensor:
    """``HybridFFN.forward`` in eval mode (``snn_ffn.py:130-145``)."""
    mlp = F.linear(F.gelu(F.linear(x, p["mlp.0.weight"], p["mlp.0.bias"])),
                   p["mlp.2.weight"], p["mlp.2.bias"])
    snn = snnffn_forward(x, {k[len("snn."):]: v for k, v in p.items() if k.startswith("snn.")},
                         T=T, L=L)
    g = torch.sigmoid(p["gate"])
    return (1 - g) * mlp + g * snn


def addition_linear(x: torch.Tensor, weight_patterns: torch.Tensor,
                    bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``AdditionLinear.forward`` (``src/maths/addition_linear.py:42-67``): -||w - x||_1."""
    diff = x.unsqueeze(1) - weight_patterns.unsqueeze(0)
    out = -torch.sum(torch.abs(diff), dim=2)
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------------------
# Episodic bank (HippocampalFormation hot path)
# --------------------------------------------------------------------------------------


class OracleBank:
    """Functional restatement of the bank part of ``HippocampalFormation``
    (``src/core/hippocampal.py:84-118,195-377``).

    Defects of the reference that this oracle (and the product) handles explicitly -- each is
    pinned in ``tests/test_oracle_vs_reference.py``:

    * full bank: ``idx = count % M`` with ``count == M`` is always slot 0 (``:200-202``).
      REPRODUCED (``overflow='reference'``).
    * candidate path: ``topk`` positions index the candidate list but are looked up as bank rows
      (``:307-317``) -> right scores, wrong ids.  FIXED here: positions are mapped through
      ``candidates``.  Scores are compared against the reference; ids against this fix.
    * candidate path: ``k = min(k, count)`` can exceed the candidate count -> ``topk`` raises
      (``:306-307``).  FIXED: ``k = min(k, n_candidates)``.
    * ``location=`` with candidates active -> shape error (``:287-289``).  FIXED: candidate rows'
      locations are used.
    * timestamps are ``time.time()`` stored in fp32 (quantised to 128 s) and ``age`` is computed
      in fp32 (``:215,296``).  REPRODUCED.
    """

    def __init__(self, max_memories=100000, feature_dim=768, spatial_dims=2,
                 use_centroid_index=True, centroids_k=256, centroids_update_interval=512):
        self.M, self.D = max_memories, feature_dim
        self.features = torch.zeros(max_memories, feature_dim)
        self.locations = torch.zeros(max_memories, spatial_dims)
        self.metadata = torch.zeros(max_memories, 4)
        self.centroids = torch.zeros(256, feature_dim)
        self.centroid_counts = torch.zeros(256)
        self.centroids_k = centroids_k
        self.centroids_update_interval = centroids_update_interval
        self.use_centroid_index = use_centroid_index
        self.index_ready = False
        self.count = 0
        self.current_location = torch.zeros(spatial_dims)
        self.idx_to_id: Dict[int, str] = {}
        self.id_to_idx: Dict[str, int] = {}

    # -- write ---------------------------------------------------------------------------
    def write(self, memory_id: str, features: torch.Tensor, now: float) -> int:
        """``create_episodic_memory`` (``hippocampal.py:195-243``); returns the slot."""
        if self.count >= self.M:
            idx = self.count % self.M
        else:
            idx = self.count
            self.count += 1
        self.features[idx] = features.detach()
        self.locations[idx] = self.current_location
        self.metadata[idx] = torch.tensor([1.0, now, 0.0, 0.0])
        if self.use_centroid_index and self.index_ready:
            eff_k = min(self.centroids_k, self.centroids.shape[0])
            dists = torch.norm(self.centroids[:eff_k] - features, dim=1)
            cidx = torch.argmin(dists)
            self.centroid_counts[cidx] += 1
            eta = 1.0 / self.centroid_counts[cidx].clamp(min=1.0)
            self.centroids[cidx] = (1 - eta) * self.centroids[cidx] + eta * features
            self.metadata[idx, 2] = cidx
        else:
            self.metadata[idx, 2] = -1
        self.id_to_idx[memory_id] = idx
        self.idx_to_id[idx] = memory_id
        if (self.use_centroid_index and self.count % self.centroids_update_interval == 0
                and self.count > self.centroids_k):
            self.rebuild_centroids()
        return idx

    # -- candidate selection ---------------------------------------------------------------
    def candidates(self, query: torch.Tensor) -> Optional[torch.Tensor]:
        """``hippocampal.py:258-270``; ``None`` means "scan the whole bank"."""
        if not (self.use_centroid_index and self.index_ready and self.count > self.centroids_k):
            return None
        c_dists = torch.norm(self.centroids - query, dim=1)
        top_c = torch.topk(-c_dists, k=min(8, self.centroids_k)).indices
        cids = self.metadata[:self.count, 2]
        mask = torch.zeros_like(cids, dtype=torch.bool)
        for cid in top_c:
            mask |= (cids == cid)
        cand = torch.nonzero(mask, as_tuple=False).squeeze(-1)
        return cand if cand.numel() > 0 else None

    # -- scoring ------------------------------------------------------------------------------
    def scores(self, query: torch.Tensor, now: float, location: Optional[torch.Tensor] = None,
               cand: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Combined score of every active (or candidate) row, ``hippocampal.py:272-303``."""
        q_norm = F.normalize(query.unsqueeze(0), dim=1)
        feats = self.features[:self.count] if cand is None else self.features[cand]
        m_norm = F.normalize(feats, dim=1)
        sim = torch.mm(q_norm, m_norm.t()).squeeze(0)
        spatial = torch.zeros_like(sim)
        if location is not None:
            locs = self.locations[:self.count] if cand is None else self.locations[cand]
            spatial = 1.0 / (1.0 + torch.norm(locs - location, dim=1))
        meta = self.metadata[:self.count] if cand is None else self.metadata[cand]
        ages = now - meta[:, 1]
        temporal = torch.exp(-ages / 3600.0)
        return (0.5 * sim + 0.3 * spatial + 0.2 * temporal) * meta[:, 0]

    def recall(self, query: torch.Tensor, k: int, now: float,
               location: Optional[torch.Tensor] = None, use_candidates: bool = True
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Top-k ``(bank_row_indices, scores)`` of ``retrieve_similar_memories`` (``:245-319``)."""
        if self.count == 0:
            return torch.empty(0, dtype=torch.long), torch.empty(0)
        cand = self.candidates(query) if use_candidates else None
        combined = self.scores(query, now, location, cand)
        k = min(k, combined.numel())
        top_scores, top_pos = torch.topk(combined, k)
        rows = top_pos if cand is None else cand[top_pos]
        return rows, top_scores

    def recall_ids(self, query, k, now, location=None) -> List[Tuple[str, float]]:
        rows, scores = self.recall(query, k, now, location)
        return [(self.idx_to_id[int(r)], float(s)) for r, s in zip(rows, scores)
                if int(r) in self.idx_to_id]

    # -- maintenance ----------------------------------------------------------------------------
    def decay(self, rate: float = 0.01) -> None:
        """``hippocampal.py:321-334``."""
        if self.count:
            self.metadata[:self.count, 0] *= (1.0 - rate)

    def rebuild_centroids(self, perm: Optional[torch.Tensor] = None) -> None:
        """``hippocampal.py:345-377``.  ``perm`` overrides the ``randperm`` draw (``:354``)."""
        if self.count == 0 or not self.use_centroid_index:
            return
        active = self.features[:self.count]
        k = min(self.centroids_k, active.shape[0])
        if perm is None:
            perm = torch.randperm(active.shape[0])
        centroids = active[perm[:k]].clone()
        dists = torch.cdist(active, centroids)
        assign = torch.argmin(dists, dim=1)
        for cid in range(k):
            mask = assign == cid
            if mask.any():
                centroids[cid] = active[mask].mean(dim=0)
        self.centroids[:k] = centroids
        if k < self.centroids_k:
            self.centroids[k:] = 0
        counts = torch.zeros(self.centroids_k)
        dists = torch.cdist(active, self.centroids[:k])
        assign = torch.argmin(dists, dim=1)
        for cid in range(k):
            counts[cid] = (assign == cid).sum()
        self.centroid_counts = counts
        self.metadata[:self.count, 2] = assign.to(self.metadata.dtype)
        self.index_ready = True


def knn_exact_batch(bank: torch.Tensor, strength: torch.Tensor, timestamps: torch.Tensor,
                    queries: torch.Tensor, k: int, now: float,
                    locations: Optional[torch.Tensor] = None,
                    query_locations: Optional[torch.Tensor] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched form of the exact path of ``retrieve_similar_memories`` (one reference call per
    query row, ``memory_augmented_layer.py:113-121``).  Returns ``(idx[nq,k], score[nq,k])``.

    Written per query with the reference's op order so it is bit-identical to calling the
    reference nq times; the bank normalisation is hoisted (it does not depend on the query).
    """
    m_norm = F.normalize(bank, dim=1)
    ages = now - timestamps
    temporal = torch.exp(-ages / 3600.0)
    k = min(k, bank.shape[0])
    idx = torch.empty(queries.shape[0], k, dtype=torch.long)
    sc = torch.empty(queries.shape[0], k)
    for i in range(queries.shape[0]):
        q_norm = F.normalize(queries[i].unsqueeze(0), dim=1)
        sim = torch.mm(q_norm, m_norm.t()).squeeze(0)
        spatial = torch.zeros_like(sim)
        if query_locations is not None:
            spatial = 1.0 / (1.0 + torch.norm(locations - query_locations[i], dim=1))
        combined = (0.5 * sim + 0.3 * spatial + 0.2 * temporal) * strength
        s, p = torch.topk(combined, k)
        idx[i], sc[i] = p, s
    return idx, sc


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard top-k lists ``[nq, S*k]`` into the global top-k; ties -> lower index.

    [build-side] The reference has no sharding; this defines the merge used by section 8e."""
    order = torch.argsort(idx, dim=1, stable=True)
    s2 = torch.gather(scores, 1, order)
    i2 = torch.gather(idx, 1, order)
    order2 = torch.argsort(s2, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(s2, 1, order2), torch.gather(i2, 1, order2)


# --------------------------------------------------------------------------------------
# Differentiable restatements (surrogate gradients) and the prosody-modulated GIF
# --------------------------------------------------------------------------------------


class MultiBitSurrogateFn(torch.autograd.Function):
    """floor/clamp forward, triangular surrogate backward: ``gif_neuron.py:6-22``."""

    @staticmethod
    def forward(ctx, inp, L):
        ctx.save_for_backward(inp)
        ctx.L = L
        return torch.clamp(torch.floor(inp), 0, L)

    @staticmethod
    def backward(ctx, grad_output):
        inp, = ctx.saved_tensors
        dist = torch.abs(inp - torch.round(inp))
        scale = torch.clamp(1.0 - 2.0 * dist, 0.0, 1.0)
        in_range = (inp >= 0.0) & (inp <= ctx.L + 1.0)
        return grad_output * in_range.float() * scale, None


class LearnableSurrogateFn(torch.autograd.Function):
    """Heaviside forward, fast-sigmoid backward incl. the slope gradient: ``neuron.py:70-108``."""

    @staticmethod
    def forward(ctx, inp, slope):
        ctx.save_for_backward(inp, slope)
        return (inp > 0).to(inp.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        inp, slope = ctx.saved_tensors
        grad_input = grad_output * (slope / ((slope * inp).abs() + 1.0) ** 2)
        raw = -grad_output * inp.abs() * inp.sign() / ((slope * inp.abs() + 1.0) ** 2)
        extra = raw.ndim - slope.ndim
        grad_slope = raw.sum(dim=list(range(extra))) if (slope.shape != raw.shape and extra > 0) else raw
        return grad_input, grad_slope


def gif_run_grad(h, v, theta, decay, L, alpha, threshold):
    """``gif_run`` with autograd through the surrogate (fp32): the graph the reference builds in
    ``gif_neuron.py:54-69`` when its inputs require grad."""
    out = []
    for t in range(h.shape[1]):
        v = v * decay + h[:, t, :]
        cl = L * theta * 2.0
        v = torch.clamp(v, -cl, cl)
        spike = MultiBitSurrogateFn.apply(v / (theta + 1e-6), L)
        v = v - spike * theta
        if alpha > 0:
            theta = theta + alpha * spike - alpha * (theta - threshold)
        out.append(spike)
    return torch.stack(out, dim=1), v, theta


def lif_step_grad(x, mem, beta, threshold, slope):
    """``VectorizedLIFNeuron.forward`` with the learnable surrogate (``neuron.py:135-139``)."""
    mem = beta * mem + x
    spk = LearnableSurrogateFn.apply(mem - threshold, slope)
    return spk, mem - spk * threshold


def prosody_gif_run(h, v, theta, gains, decay, L, alpha, threshold, strength):
    """``ProsodyModulatedGIF.forward`` loop (``prosody_gif.py:69-106``): attention gains [rows, T]
    scale the input, the effective threshold (clamped to [0.5, 1.5] x) and the adaptation rate.
    Note: no 1e-6 in the division, unlike ``GIFNeuron``."""
    out = []
    for t in range(h.shape[1]):
        i_t = h[:, t, :]
        if gains is not None:
            g = gains[:, t].unsqueeze(1)
            i_t = i_t * g
        v = v * decay + i_t
        th_eff = theta
        if gains is not None:
            scale = torch.clamp(1.0 - strength * (g - 1.0), 0.5, 1.5)
            th_eff = theta * scale
        cl = L * th_eff * 2.0
        v = torch.clamp(v, -cl, cl)
        spike = torch.clamp(torch.floor(v / th_eff), 0, L)
        v = v - spike * th_eff
        if alpha > 0:
            a_eff = alpha * g if gains is not None else alpha
            theta = theta + a_eff * spike - a_eff * (theta - threshold)
        out.append(spike)
    return torch.stack(out, dim=1), v, theta


def prosody_gif_run_grad(h, v, theta, gains, decay, L, alpha, threshold, strength):
    """``prosody_gif_run`` with autograd through ``MultiBitSurrogate``: the graph ``ProsodyModulatedGIF.forward``
    builds (``prosody_gif.py:64-106``) when its inputs require grad (fp32)."""
    out = []
    for t in range(h.shape[1]):
        i_t = h[:, t, :]
        if gains is not None:
            g = gains[:, t].unsqueeze(1)
            i_t = i_t * g
        v = v * decay + i_t
        th_eff = theta
        if gains is not None:
            scale = torch.clamp(1.0 - strength * (g - 1.0), 0.5, 1.5)
            th_eff = theta * scale
        cl = L * th_eff * 2.0
        v = torch.clamp(v, -cl, cl)
        spike = MultiBitSurrogateFn.apply(v / th_eff, L)
        v = v - spike * th_eff
        if alpha > 0:
            a_eff = alpha * g if gains is not None else alpha
            theta = theta + a_eff * spike - a_eff * (theta - threshold)
        out.append(spike)
    return torch.stack(out, dim=1), v, theta
