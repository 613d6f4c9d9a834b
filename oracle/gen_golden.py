#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container (reference mounted read-only at /root/reference):
    python oracle/gen_golden.py
Every fixture holds seeded inputs and the outputs of the unmodified reference modules
(loaded through oracle/_ref_loader.py).  The fixtures are data only -- no reference source
travels.  tests/test_golden.py checks the oracle (CPU) and the HIP path (GPU) against them.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from oracle import _ref_loader as L

OUT = os.path.join(ROOT, "tests", "golden")
NOW = 1.7e9 + 12345.678


def save(name, obj):
    path = os.path.join(OUT, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    N = L.load("base.neuron")
    G = L.load("src.core.language_zone.gif_neuron")
    F = L.load("src.core.language_zone.snn_ffn")
    H = L.load("core.hippocampal")
    Z = L.load("src.base.snn_brain_zones")
    H.time.time = lambda: NOW

    # --- Izhikevich, BASELINE config 1: 256 neurons x 100 steps, I = 20*rand, RS params ----------
    torch.manual_seed(0)
    I = 20 * torch.rand(256, 100)
    izh = N.IzhikevichNeuron(0.02, 0.2, -65.0, 8.0, 0.2)
    s1 = izh(I); v1, u1 = izh.v.clone(), izh.u.clone()
    s2 = izh(I)
    I3 = 20 * torch.rand(3, 40, 20)
    izh3 = N.IzhikevichNeuron(0.02, 0.2, -65.0, 8.0, 0.2)
    s3 = izh3(I3)
    save("izhikevich.pt", dict(params=(0.02, 0.2, -65.0, 8.0, 0.2), I=I, spikes=s1.to(torch.uint8), v=v1, u=u1,
                               spikes_second_call=s2.to(torch.uint8), v2=izh.v.clone(), u2=izh.u.clone(),
                               I3=I3, spikes3=s3.contiguous().to(torch.uint8), v3=izh3.v.clone()))

    # --- AdEx -------------------------------------------------------------------------------------
    torch.manual_seed(1)
    Ia = 600 * torch.rand(64, 120)
    ad = N.AdExNeuron(a=2.0, b=60.0)
    sa = ad(Ia)
    save("adex.pt", dict(kwargs=dict(a=2.0, b=60.0), I=Ia, spikes=sa.to(torch.uint8), V=ad.V.clone(), w=ad.w.clone()))

    # --- LIF: 6 consecutive steps -------------------------------------------------------------------
    torch.manual_seed(2)
    lif = N.VectorizedLIFNeuron(48, beta=0.95, threshold=0.5)
    xs = torch.randn(6, 5, 48)
    spk, mem = [], []
    for t in range(6):
        s, m = lif(xs[t]); spk.append(s.detach().clone()); mem.append(m.detach().clone())
    save("lif.pt", dict(beta=0.95, threshold=0.5, x=xs, spikes=torch.stack(spk).to(torch.uint8), mem=torch.stack(mem)))

    # --- GIF fp32 and bf16 (whole module: linear + loop) -------------------------------------------------
    gif = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(3)
        g = G.GIFNeuron(48, 96, L=8).to(dt)
        x = (torch.randn(6, 16, 48) * 3).to(dt)
        with torch.no_grad():
            h = g.linear(x)
            s, (v, th) = g(x)
        gif[name] = dict(x=x, h=h, weight=g.linear.weight.detach().clone(), bias=g.linear.bias.detach().clone(),
                         spikes=s, v=v, theta=th, decay=g.decay, L=8, alpha=g.alpha, threshold=g.threshold)
    save("gif.pt", gif)

    # --- SNNFFN / HybridFFN (eval) ------------------------------------------------------------------------
    torch.manual_seed(4)
    ffn = F.SNNFFN(64, 256, num_timesteps=4, L=8).eval()
    hy = F.HybridFFN(64, 256, num_timesteps=4, L=8).eval()
    x = torch.randn(2, 16, 64)
    with torch.no_grad():
        y, yh = ffn(x), hy(x)
    save("snnffn.pt", dict(x=x, ffn_state={k: v.clone() for k, v in ffn.state_dict().items()}, ffn_out=y,
                           hybrid_state={k: v.clone() for k, v in hy.state_dict().items()}, hybrid_out=yh, T=4, L=8))

    # --- episodic bank scenario: 600 writes with online centroids + rebuilds, recall ------------------------
    D, M = 32, 2000
    hf = H.HippocampalFormation(n_place_cells=10, n_time_cells=5, n_grid_cells=5, max_memories=M,
                                feature_dim=D, device="cpu")
    hf.centroids_k = 16
    hf.centroids_update_interval = 64
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(600, D, generator=g) * (2 * torch.rand(600, 1, generator=g))
    torch.manual_seed(5)                                   # drives the rebuilds' randperm
    for i in range(600):
        hf.create_episodic_memory(f"m{i}", f"e{i}", feats[i])
    q = feats[[7, 100, 333, 599]] + 0.05 * torch.randn(4, D, generator=g)
    hf.use_centroid_index = False
    exact = [hf.retrieve_similar_memories(q[j], k=10) for j in range(4)]
    loc = torch.tensor([0.3, -0.2])
    with_loc = [hf.retrieve_similar_memories(q[j], location=loc, k=10) for j in range(4)]
    meta_before_decay = hf.memory_metadata[:600].clone()
    hf.decay_memories(0.1)
    after_decay = [hf.retrieve_similar_memories(q[j], k=10) for j in range(4)]
    hf.use_centroid_index = True
    cand_scores = [[s for _, s in hf.retrieve_similar_memories(q[j], k=5)] for j in range(4)]  # ids are wrong upstream
    save("bank.pt", dict(D=D, M=M, now=NOW, centroids_k=16, interval=64, seed=5, feats=feats, queries=q, loc=loc,
                         metadata=meta_before_decay, centroids=hf.centroids.clone(),
                         centroid_counts=hf.centroid_counts.clone(), exact=exact, with_loc=with_loc,
                         after_decay=after_decay, candidate_scores=cand_scores))

    # --- brain zone (AdditionLinear -> izh + lif groups -> AdditionLinear) ------------------------------------
    torch.manual_seed(6)
    cfgs = [Z.SpikingNeuronConfig("izh_rs", "s", "glu", 50.0, a=0.02, b=0.2, c=-65.0, d=8.0, dt=0.2),
            Z.SpikingNeuronConfig("lif", "s", "glu", 50.0, threshold=0.5, beta_decay=0.95)]
    zone = Z.NeuromorphicBrainZone(Z.BrainZoneConfig(name="z", max_neurons=64, d_model=32, spiking_configs=cfgs))
    xz = torch.randn(6, 32)
    with torch.no_grad():
        out, info = zone(xz)
    save("zone.pt", dict(x=xz, state={k: v.clone() for k, v in zone.state_dict().items()}, out=out,
                         avg_firing_rate=info["avg_firing_rate"]))

    # --- surrogate-gradient training: GIF BPTT (upstream gradients through MultiBitSurrogate) -------------
    torch.manual_seed(7)
    g = G.GIFNeuron(20, 36, L=8, alpha=0.05)
    x = (torch.randn(5, 12, 20) * 3).requires_grad_(True)
    v0 = (0.3 * torch.randn(5, 36)).requires_grad_(True)
    th0 = (1.0 + 0.2 * torch.rand(5, 36)).requires_grad_(True)
    s, (v, th) = g(x, state=(v0, th0))
    h = g.linear(x).detach()
    ws, wv, wt = torch.randn_like(s), torch.randn_like(v), torch.randn_like(th)
    loss = (s * ws).sum() + (v * wv).sum() + (th * wt).sum()
    gx, gv0, gth0, gW, gb = torch.autograd.grad(loss, [x, v0, th0, g.linear.weight, g.linear.bias])
    save("gif_grad.pt", dict(x=x.detach(), h=h, v0=v0.detach(), theta0=th0.detach(),
                             weight=g.linear.weight.detach().clone(), bias=g.linear.bias.detach().clone(),
                             decay=g.decay, L=8, alpha=0.05, threshold=g.threshold,
                             spikes=s.detach(), v=v.detach(), theta=th.detach(), w_spikes=ws, w_v=wv, w_theta=wt,
                             g_x=gx, g_v0=gv0, g_theta0=gth0, g_weight=gW, g_bias=gb))

    # --- LIF with the learnable surrogate: 4 chained steps, gradients to inputs and slope ----------------
    torch.manual_seed(8)
    lif = N.VectorizedLIFNeuron(40, beta=0.9, threshold=0.5, init_slope=5.0)
    xs = (0.6 * torch.randn(4, 3, 40)).requires_grad_(True)
    outs = [lif(xs[t]) for t in range(4)]
    wl = torch.randn(4, 2, 3, 40)
    loss = sum((outs[t][0] * wl[t, 0]).sum() for t in range(4)) + (outs[3][1] * wl[3, 1]).sum()
    gxs, gslope = torch.autograd.grad(loss, [xs, lif.slope])
    save("lif_grad.pt", dict(x=xs.detach(), beta=0.9, threshold=0.5, slope=5.0, w=wl,
                             spikes=torch.stack([o[0].detach() for o in outs]), mem=outs[3][1].detach(),
                             g_x=gxs, g_slope=gslope))

    # --- ProsodyModulatedGIF (forward) ---------------------------------------------------------------------
    PG = L.load("src.core.language_zone.prosody_gif")
    torch.manual_seed(9)
    pg = PG.ProsodyModulatedGIF(24, 48, L=8, alpha=0.05, attention_modulation_strength=0.3)
    x = torch.randn(4, 10, 24) * 3
    gains = 0.5 + 2.5 * torch.rand(4, 10)
    with torch.no_grad():
        s1, (v1, t1) = pg(x, attention_gains=gains)
        s2, (v2, t2) = pg(x, attention_gains=gains, state=(v1, t1))
        s3, (v3, t3) = pg(x)
        h = pg.linear(x)
    save("prosody_gif.pt", dict(x=x, h=h, gains=gains, weight=pg.linear.weight.detach().clone(),
                                bias=pg.linear.bias.detach().clone(), decay=pg.decay, L=8, alpha=0.05,
                                threshold=pg.threshold, strength=0.3, spikes=s1, v=v1, theta=t1,
                                spikes_cont=s2, v_cont=v2, theta_cont=t2, spikes_nogain=s3, v_nogain=v3,
                                theta_nogain=t3))


if __name__ == "__main__":
    main()
