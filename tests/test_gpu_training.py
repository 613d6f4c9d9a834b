"""GPU parity for SURVEY.md section 8f-4: surrogate-gradient backward kernels (GIF BPTT, LIF step)
and the prosody-modulated GIF, against the CPU oracle's autograd restatement (itself bit-equal to
the reference's autograd, tests/test_oracle_vs_reference.py).  Forward results are bit-exact;
gradients within the north-star tolerance |a - b| <= 1e-5 * max(1, |b|)."""
import math

import pytest
import torch

from oracle import aura_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _close(a, b, tol=TOL):
    a, b = a.detach().cpu(), b.detach()
    bad = (a - b).abs() > tol * b.abs().clamp_min(1.0)
    return not bool(bad.any())


@pytest.mark.parametrize("rows,T,H,L,alpha", [(7, 12, 64, 8, 0.01), (3, 5, 37, 4, 0.05), (64, 16, 256, 8, 0.0),
                                              (1, 1, 4, 16, 0.01), (33, 40, 100, 2, 0.2)])
def test_gif_bptt_matches_oracle(dev, rows, T, H, L, alpha):
    from aura_snn_rag_amd.core.language_zone.gif_neuron import run_gif_loop, run_gif_loop_grad
    g = torch.Generator().manual_seed(rows * 131 + T * 7 + H)
    decay, thr0 = math.exp(-0.1), 1.0
    h = torch.randn(rows, T, H, generator=g) * 3
    v0 = 0.4 * torch.randn(rows, H, generator=g)
    t0 = 1.0 + 0.3 * torch.rand(rows, H, generator=g)
    ws, wv, wt = (torch.randn(s, generator=g) for s in ((rows, T, H), (rows, H), (rows, H)))
    # oracle
    ho, vo, to = (x.clone().requires_grad_(True) for x in (h, v0, t0))
    s, v, th = O.gif_run_grad(ho, vo, to, decay, L, alpha, thr0)
    ref = torch.autograd.grad((s * ws).sum() + (v * wv).sum() + (th * wt).sum(), [ho, vo, to])
    # HIP
    hd, vd, td = (x.to(dev).requires_grad_(True) for x in (h, v0, t0))
    sd, (vT, tT) = run_gif_loop_grad(hd, (vd, td), decay=decay, L=L, alpha=alpha, threshold=thr0)
    assert torch.equal(sd.cpu(), s.detach()) and torch.equal(vT.cpu(), v.detach()) and torch.equal(tT.cpu(), th.detach())
    got = torch.autograd.grad((sd * ws.to(dev)).sum() + (vT * wv.to(dev)).sum() + (tT * wt.to(dev)).sum(),
                              [hd, vd, td])
    for a, b, name in zip(got, ref, ("g_h", "g_v0", "g_theta0")):
        assert _close(a, b), f"{name}: max err {(a.cpu() - b).abs().max().item():.3e}"
    assert any(r.abs().sum() > 0 for r in ref)
    # the training forward is the inference forward
    with torch.no_grad():
        s2, (v2, t2) = run_gif_loop(h.to(dev), (v0.to(dev), t0.to(dev)), decay=decay, L=L, alpha=alpha,
                                    threshold=thr0, T=T)
    assert torch.equal(s2, sd) and torch.equal(v2, vT) and torch.equal(t2, tT)


def test_gif_clamp_routes_gradient_to_theta(dev):
    """Values beyond +-2*L*theta are clamped: their gradient goes to theta through the tensor bounds
    (torch.clamp semantics), not to the input."""
    from aura_snn_rag_amd.core.language_zone.gif_neuron import run_gif_loop_grad
    h = torch.tensor([[[100.0, -100.0, 1.5, 0.0]]])
    ho = h.clone().requires_grad_(True)
    to = torch.ones(1, 4, requires_grad=True)
    s, v, th = O.gif_run_grad(ho, torch.zeros(1, 4), to, 1.0, 2, 0.0, 1.0)
    ref = torch.autograd.grad(v.sum() + s.sum(), [ho, to])
    hd = h.to(dev).requires_grad_(True)
    td = torch.ones(1, 4, device=dev, requires_grad=True)
    sd, (vd, _) = run_gif_loop_grad(hd, (torch.zeros(1, 4, device=dev), td), decay=1.0, L=2, alpha=0.0,
                                    threshold=1.0)
    got = torch.autograd.grad(vd.sum() + sd.sum(), [hd, td])
    assert torch.equal(got[0].cpu(), ref[0]) and _close(got[1], ref[1])
    assert ref[0][0, 0, 0] == 0 and ref[0][0, 0, 1] == 0           # clamped lanes: no input gradient


def test_gif_module_training_step(dev):
    """GIFNeuron / BalancedGIFNeuron record history when parameters require grad; weight gradients
    flow through the library GEMM's autograd and match the oracle fed the same currents."""
    from aura_snn_rag_amd.core.language_zone.gif_neuron import BalancedGIFNeuron, GIFNeuron
    torch.manual_seed(0)
    for cls in (GIFNeuron, BalancedGIFNeuron):
        n = cls(32, 64, L=8, alpha=0.02).to(dev)
        x = torch.randn(4, 10, 32, device=dev) * 3
        w = torch.randn(4, 10, 64, device=dev)
        cur = n.currents(x)
        cur.retain_grad()
        from aura_snn_rag_amd.core.language_zone.gif_neuron import run_gif_loop_grad
        s, _ = run_gif_loop_grad(cur, None, decay=n.decay, L=n.L, alpha=n.alpha, threshold=n.threshold)
        (s * w).sum().backward()
        gh = cur.grad.clone()
        grads_direct = [None if p.grad is None else p.grad.clone() for p in n.parameters()]
        n.zero_grad()
        s2, (v2, t2) = n(x)
        assert s2.requires_grad and torch.equal(s2, s)
        (s2 * w).sum().backward()
        for p, gd in zip(n.parameters(), grads_direct):     # BalancedGIF's inherited `linear` is unused
            assert (p.grad is None and gd is None) or torch.equal(p.grad, gd)
        co = cur.detach().cpu().requires_grad_(True)
        so, _, _ = O.gif_run_grad(co, torch.zeros(4, 64), torch.full((4, 64), float(n.threshold)), n.decay,
                                  n.L, n.alpha, n.threshold)
        (so * w.cpu()).sum().backward()
        assert torch.equal(so.detach(), s.cpu()) and _close(gh, co.grad)
        with torch.no_grad():
            s3, _ = n(x)
        assert not s3.requires_grad and torch.equal(s3, s)
    # bf16 parameters record autograd too since round 2 (test_gif_bptt_bf16); other dtypes are refused
    sb, _ = GIFNeuron(8, 8).to(dev).to(torch.bfloat16)(torch.randn(1, 2, 8, device=dev, dtype=torch.bfloat16))
    assert sb.requires_grad and sb.dtype == torch.bfloat16


def test_snnffn_training_path(dev):
    """Training-mode SNNFFN: same forward as the fused inference path, finite gradients everywhere,
    and input gradients equal to the oracle's when it is fed the GPU's own GEMM outputs."""
    from aura_snn_rag_amd.core.language_zone.snn_ffn import HybridFFN, SNNFFN
    torch.manual_seed(1)
    ffn = SNNFFN(64, 128, num_timesteps=4, L=8, dropout=0.0).to(dev)
    x = torch.randn(2, 12, 64, device=dev, requires_grad=True)
    y = ffn(x)
    with torch.no_grad():
        y0 = ffn(x)
    assert y.requires_grad and torch.allclose(y, y0, atol=1e-6, rtol=0)
    y.square().sum().backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    for name, p in ffn.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert ffn.syn1.weight.grad.abs().sum() > 0 and ffn.neuron2.linear.weight.grad.abs().sum() > 0
    hy = HybridFFN(64, 128, num_timesteps=4, L=8, dropout=0.0).to(dev)
    hy(x.detach()).sum().backward()
    assert hy.gate.grad is not None and torch.isfinite(hy.gate.grad)


@pytest.mark.parametrize("shape,size", [((5, 48), 48), ((2, 3, 33), 33), ((1, 4), 4), ((64, 1024), 1024)])
def test_lif_surrogate_gradients(dev, shape, size):
    from aura_snn_rag_amd.base.neuron import VectorizedLIFNeuron
    g = torch.Generator().manual_seed(size)
    lif = VectorizedLIFNeuron(size, beta=0.9, threshold=0.6, init_slope=4.0).to(dev)
    with torch.no_grad():
        lif.slope.copy_((2 + 6 * torch.rand(size, generator=g)).to(dev))
    xs = [torch.randn(shape, generator=g) for _ in range(3)]
    ws = [torch.randn(shape, generator=g) for _ in range(4)]
    xo = [x.clone().requires_grad_(True) for x in xs]
    slope = lif.slope.detach().cpu().clone().requires_grad_(True)
    beta, thr = lif.beta.cpu(), lif.threshold.cpu()
    mem, loss = torch.zeros(shape), 0.0
    spikes_ref = []
    for t in range(3):
        spk, mem = O.lif_step_grad(xo[t], mem, beta, thr, slope)
        spikes_ref.append(spk.detach())
        loss = loss + (spk * ws[t]).sum()
    loss = loss + (mem * ws[3]).sum()
    ref = torch.autograd.grad(loss, xo + [slope])
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    loss = 0.0
    for t in range(3):
        spk, m = lif(xd[t])
        assert torch.equal(spk.detach().cpu(), spikes_ref[t])
        loss = loss + (spk * ws[t].to(dev)).sum()
    assert torch.equal(m.detach().cpu(), mem.detach())
    loss = loss + (m * ws[3].to(dev)).sum()
    got = torch.autograd.grad(loss, xd + [lif.slope])
    for a, b in zip(got[:3], ref[:3]):
        assert _close(a, b)
    # slope gradient is a sum over the batch: tolerance relative to the summed magnitude
    assert torch.allclose(got[3].cpu(), ref[3], rtol=1e-4, atol=1e-5 * max(1.0, ref[3].abs().max().item()))
    # reset + no-grad call go back to the in-place inference kernel
    lif.reset_mem()
    with torch.no_grad():
        s0, m0 = lif(xs[0].to(dev))
    assert not s0.requires_grad and torch.equal(s0.cpu(), spikes_ref[0])


@pytest.mark.parametrize("B,T,I,H", [(4, 10, 24, 48), (3, 7, 9, 21), (16, 32, 64, 256)])
def test_prosody_gif_module(dev, B, T, I, H):
    from aura_snn_rag_amd.core.language_zone.prosody_gif import ProsodyModulatedGIF
    torch.manual_seed(B + T)
    pg = ProsodyModulatedGIF(I, H, L=8, alpha=0.05, attention_modulation_strength=0.3).to(dev)
    x = torch.randn(B, T, I, device=dev) * 3
    gains = (0.2 + 3.0 * torch.rand(B, T)).to(dev)
    with torch.no_grad():
        h = pg.linear(x).cpu()
        for gn in (gains, None):
            s, (v, th) = pg(x, attention_gains=gn)
            rs, rv, rt = O.prosody_gif_run(h, torch.zeros(B, H), torch.full((B, H), pg.threshold),
                                           None if gn is None else gn.cpu(), pg.decay, 8, 0.05, pg.threshold, 0.3)
            assert torch.equal(s.cpu(), rs) and torch.equal(v.cpu(), rv) and torch.equal(th.cpu(), rt)
            s2, (v2, t2) = pg(x, attention_gains=gn, state=(v, th))
            rs2, rv2, rt2 = O.prosody_gif_run(h, rv, rt, None if gn is None else gn.cpu(), pg.decay, 8, 0.05,
                                              pg.threshold, 0.3)
            assert torch.equal(s2.cpu(), rs2) and torch.equal(v2.cpu(), rv2) and torch.equal(t2.cpu(), rt2)
    # surrogate-gradient backward (aura_gif_prosody_backward) against the oracle's autograd restatement, which
    # is bit-equal to the reference's own graph (tests/test_oracle_vs_reference.py): input, gains, carried state
    # and the linear layer
    for use_gains in (True, False):
        xg = (torch.randn(B, T, I) * 3)
        gn = (0.2 + 3.0 * torch.rand(B, T)) if use_gains else None
        v0 = 0.5 * torch.randn(B, H)
        t0 = 1 + 0.3 * torch.rand(B, H)
        ws = torch.randn(B, T, H)
        cpu = [t.clone().requires_grad_(True) for t in (xg, v0, t0)] + ([gn.clone().requires_grad_(True)] if use_gains else [])
        w_cpu = pg.linear.weight.detach().cpu().clone().requires_grad_(True)
        b_cpu = pg.linear.bias.detach().cpu().clone()
        os_, ov, ot = O.prosody_gif_run_grad(torch.nn.functional.linear(cpu[0], w_cpu, b_cpu), cpu[1], cpu[2],
                                             cpu[3] if use_gains else None, pg.decay, 8, 0.05, pg.threshold, 0.3)
        ref = torch.autograd.grad((os_ * ws).sum() + ov.sum() - ot.sum(), cpu + [w_cpu])
        gpu = [t.to(dev).requires_grad_(True) for t in (xg, v0, t0)] + ([gn.to(dev).requires_grad_(True)] if use_gains else [])
        s, (v, th) = pg(gpu[0], state=(gpu[1], gpu[2]), attention_gains=gpu[3] if use_gains else None)
        got = torch.autograd.grad((s * ws.to(dev)).sum() + v.sum() - th.sum(), gpu + [pg.linear.weight])
        # the spikes may differ where the GPU's GEMM rounds a current differently: compare gradients only if they agree
        if torch.equal(s.detach().cpu(), os_.detach()):
            for a, b in zip(got, ref):
                assert torch.allclose(a.cpu(), b, rtol=2e-4, atol=2e-4 * max(1.0, float(b.abs().max()))), \
                    float((a.cpu() - b).abs().max())
    with pytest.raises(ValueError):
        with torch.no_grad():
            pg(x, attention_gains=gains[:, :-1])


def _rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _gif_bf16_bptt_fp32_math(h, v0, t0, ws, wv, wt, decay, L, alpha, thr0):
    """What aura_gif_backward_bf16 is meant to compute, in torch on the CPU: the bf16 forward (per-op rounding),
    then BPTT in fp32 from the saved bf16 (a_t, theta_{t-1}) with the forward's roundings re-applied."""
    rows, T, H = h.shape
    f = lambda x: x.to(torch.float32)
    v, th = f(v0), f(t0)
    A, TH = [], []
    for t in range(T):
        a = _rb(_rb(v * decay) + f(h[:, t]))
        cl = _rb(_rb(L * th) * 2.0)
        b = torch.minimum(torch.maximum(a, -cl), cl)
        d = _rb(th + 1e-6)
        n = _rb(b / d)
        s = torch.clamp(torch.floor(n), 0, L)
        A.append(a); TH.append(th)
        v = _rb(b - _rb(s * th))
        if alpha > 0:
            th = _rb(_rb(th + _rb(alpha * s)) - _rb(alpha * _rb(th - thr0)))
    gv, gth = f(wv).clone(), f(wt).clone()
    gh = torch.zeros(rows, T, H)
    for t in range(T - 1, -1, -1):
        a, thp = A[t], TH[t]
        cl = _rb(_rb(L * thp) * 2.0)
        b = torch.minimum(torch.maximum(a, -cl), cl)
        d = _rb(thp + 1e-6)
        n = _rb(b / d)
        s = torch.clamp(torch.floor(n), 0, L)
        gs = f(ws[:, t]).clone()
        gthp = gth.clone()
        if alpha > 0:
            gs = gs + alpha * gth
            gthp = gthp - alpha * gth
        gs = gs - thp * gv
        gthp = gthp - s * gv
        gb = gv.clone()
        tri = torch.clamp(1.0 - 2.0 * (n - torch.round(n)).abs(), 0.0, 1.0)
        sur = torch.where((n >= 0) & (n <= L + 1.0), tri, torch.zeros_like(tri))
        gn = gs * sur
        gb = gb + gn / d
        gthp = gthp + (-gn * b / (d * d))
        lo, hi = a < -cl, a > cl
        ga = torch.where(lo | hi, torch.zeros_like(gb), gb)
        gcl = torch.where(lo, -gb, torch.where(hi, gb, torch.zeros_like(gb)))
        gthp = gthp + gcl * (2.0 * L)
        gh[:, t] = ga
        gv = ga * decay
        gth = gthp
    return gh, gv, gth


@pytest.mark.parametrize("rows,T,H,L,alpha", [(7, 12, 64, 8, 0.01), (3, 5, 37, 4, 0.05), (32, 16, 256, 8, 0.0),
                                              (16, 16, 128, 16, 0.02)])
def test_gif_bptt_bf16(dev, rows, T, H, L, alpha):
    """bf16 training path of the GIF loop: the recording forward is the per-op-rounded bf16
    inference forward bit for bit; the gradients equal the intended arithmetic (fp32 BPTT from the saved bf16
    values) to one bf16 rounding, and sit within bf16 noise of the reference's own bf16 autograd graph
    (oracle restatement of gif_neuron.py:54-69 + MultiBitSurrogate on bf16 CPU tensors)."""
    from aura_snn_rag_amd.core.language_zone.gif_neuron import run_gif_loop, run_gif_loop_grad
    g = torch.Generator().manual_seed(rows * 31 + T * 5 + H)
    decay, thr0 = math.exp(-0.1), 1.0
    bf = torch.bfloat16
    h = (torch.randn(rows, T, H, generator=g) * 3).to(bf)
    v0 = (0.4 * torch.randn(rows, H, generator=g)).to(bf)
    t0 = (1.0 + 0.3 * torch.rand(rows, H, generator=g)).to(bf)
    ws, wv, wt = (torch.randn(s, generator=g).to(bf) for s in ((rows, T, H), (rows, H), (rows, H)))
    # HIP
    hd, vd, td = (x.to(dev).requires_grad_(True) for x in (h, v0, t0))
    sd, (vT, tT) = run_gif_loop_grad(hd, (vd, td), decay=decay, L=L, alpha=alpha, threshold=thr0)
    assert sd.dtype == bf and vT.dtype == bf
    with torch.no_grad():
        s2, (v2, t2) = run_gif_loop(h.to(dev), (v0.to(dev), t0.to(dev)), decay=decay, L=L, alpha=alpha,
                                    threshold=thr0, T=T)
    assert torch.equal(s2, sd) and torch.equal(v2, vT) and torch.equal(t2, tT)
    got = torch.autograd.grad((sd * ws.to(dev)).sum() + (vT * wv.to(dev)).sum() + (tT * wt.to(dev)).sum(),
                              [hd, vd, td])
    assert all(x.dtype == bf for x in got)
    # (1) the intended arithmetic, to one bf16 rounding of the result
    want = _gif_bf16_bptt_fp32_math(h, v0, t0, ws, wv, wt, decay, L, alpha, thr0)
    for a, b, name in zip(got, want, ("g_h", "g_v0", "g_theta0")):
        a32 = a.float().cpu()
        err = (a32 - b).abs()
        tol = 2.0 ** -7 * b.abs() + 1e-6
        frac = float((err <= tol).float().mean())
        assert frac > 0.999, f"{name}: {1 - frac:.2e} of the entries beyond one bf16 rounding (max err {err.max():.3e})"
    # (2) the reference's bf16 autograd graph: same spikes/state, gradients within bf16 noise
    ho, vo, to = (x.clone().requires_grad_(True) for x in (h, v0, t0))
    s, v, th = O.gif_run_grad(ho, vo, to, decay, L, alpha, thr0)
    assert torch.equal(sd.cpu(), s.detach()) and torch.equal(vT.cpu(), v.detach()) and torch.equal(tT.cpu(), th.detach())
    ref = torch.autograd.grad((s * ws).sum() + (v * wv).sum() + (th * wt).sum(), [ho, vo, to])
    for a, b, name in zip(got, ref, ("g_h", "g_v0", "g_theta0")):
        a32, b32 = a.float().cpu().flatten(), b.float().flatten()
        if float(b32.norm()) == 0.0:
            assert float(a32.norm()) == 0.0
            continue
        cos = float(torch.dot(a32, b32) / (a32.norm() * b32.norm()))
        rel = float((a32 - b32).norm() / b32.norm())
        assert cos > 0.995 and rel < 0.08, f"{name}: cosine {cos:.5f}, relative L2 distance {rel:.4f} to the bf16 autograd"


def test_gif_module_trains_in_bf16(dev):
    """GIFNeuron and SNNFFN record autograd for bf16 inputs (they raised NotImplementedError in round 1)."""
    from aura_snn_rag_amd.core.language_zone.gif_neuron import GIFNeuron
    from aura_snn_rag_amd.core.language_zone.snn_ffn import SNNFFN
    torch.manual_seed(0)
    n = GIFNeuron(32, 64, L=8).to(dev).to(torch.bfloat16)
    x = torch.randn(4, 6, 32, device=dev, dtype=torch.bfloat16, requires_grad=True)
    s, (v, th) = n(x)
    (s.float().mean() + v.float().mean()).backward()
    assert x.grad is not None and x.grad.dtype == torch.bfloat16 and bool(torch.isfinite(x.grad.float()).all())
    assert n.linear.weight.grad is not None and float(n.linear.weight.grad.float().abs().sum()) > 0
    ffn = SNNFFN(32, 64, num_timesteps=4, L=8).to(dev).to(torch.bfloat16).train()
    y = ffn(torch.randn(2, 5, 32, device=dev, dtype=torch.bfloat16, requires_grad=True))
    y.float().sum().backward()
    assert all(p.grad is not None for p in ffn.parameters() if p.requires_grad)


@pytest.mark.parametrize("B,IN,OUT,bias", [(5, 33, 17, True), (64, 256, 128, False), (130, 70, 200, True), (1, 4, 3, False)])
def test_addition_linear_gradients(dev, B, IN, OUT, bias):
    """AdditionLinear records autograd history when the reference would (it raised NotImplementedError for inputs
    that require grad and silently dropped the templates' gradient until round 3): gradients of the input, the
    templates and the bias against autograd over the oracle's restatement of addition_linear.py:50-64, exact
    ties (x == w: sign(0) = 0) included."""
    from aura_snn_rag_amd.maths.addition_linear import AdditionLinear
    g = torch.Generator().manual_seed(B * 7 + IN)
    lin = AdditionLinear(IN, OUT, bias=bias)
    x = torch.randn(B, IN, generator=g) * 0.2
    with torch.no_grad():
        x[0, : min(IN, 3)] = lin.weight_patterns[min(OUT - 1, 1), : min(IN, 3)]      # exact ties
        if bias:
            lin.bias.copy_(torch.randn(OUT, generator=g))
    wts = torch.randn(B, OUT, generator=g)
    xo = x.clone().requires_grad_(True)
    wo = lin.weight_patterns.detach().clone().requires_grad_(True)
    bo = lin.bias.detach().clone().requires_grad_(True) if bias else None
    ref_out = O.addition_linear(xo, wo, bo)
    ref = torch.autograd.grad((ref_out * wts).sum(), [xo, wo] + ([bo] if bias else []))
    lin = lin.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out = lin(xd)
    assert out.requires_grad and torch.allclose(out.detach().cpu(), ref_out.detach(), rtol=1e-5, atol=1e-4)
    got = torch.autograd.grad((out * wts.to(dev)).sum(), [xd, lin.weight_patterns] + ([lin.bias] if bias else []))
    for a, b, name in zip(got, ref, ("g_x", "g_w", "g_bias")):
        scale = max(1.0, float(b.abs().max()))
        assert torch.allclose(a.cpu(), b, rtol=1e-5, atol=2e-5 * scale), (name, float((a.cpu() - b).abs().max()))
    # only the templates require grad (a frozen input): still recorded, no input gradient computed
    out2 = lin(x.to(dev))
    assert out2.requires_grad
    (gw2,) = torch.autograd.grad((out2 * wts.to(dev)).sum(), [lin.weight_patterns])
    assert torch.equal(gw2, got[1])
    with torch.no_grad():
        assert not lin(x.to(dev)).requires_grad


def test_brain_zone_trains(dev):
    """A zone with a LIF population sends gradients to both AdditionLinear projections, the LIF surrogate's slope
    and the input (the reference's graph: addition_linear.py:50-64 + neuron.py:135-139 with the learnable
    surrogate); the out-projection's template gradient equals autograd over the oracle given the same spikes."""
    from aura_snn_rag_amd.base.snn_brain_zones import BrainZoneConfig, NeuromorphicBrainZone, SpikingNeuronConfig
    torch.manual_seed(1)
    cfgs = [SpikingNeuronConfig("lif", "s", "glu", 100.0, threshold=0.5, beta_decay=0.95)]
    zone = NeuromorphicBrainZone(BrainZoneConfig(name="z", max_neurons=48, d_model=24, spiking_configs=cfgs)).to(dev)
    with torch.no_grad():                                  # currents around the threshold: some neurons spike
        zone.input_projection.weight_patterns.mul_(0.1)
        zone.neuron_groups["lif"].homeo_i.fill_(1.5)
    x = (torch.randn(9, 24, device=dev) * 0.05).requires_grad_(True)
    wts = torch.randn(9, 24, device=dev)
    out, info = zone(x)
    assert out.requires_grad and 0.0 < info["avg_firing_rate"] < 1.0
    params = [zone.input_projection.weight_patterns, zone.output_projection.weight_patterns,
              zone.neuron_groups["lif"].core.slope]
    got = torch.autograd.grad((out * wts).sum(), [x] + params)
    assert all(t is not None and bool(torch.isfinite(t).all()) for t in got)
    assert float(got[0].abs().sum()) > 0 and float(got[1].abs().sum()) > 0 and float(got[3].abs().sum()) > 0
    # out-projection: oracle autograd given the zone's own spikes
    with torch.no_grad():
        zone.neuron_groups["lif"].core.reset_mem()
        zin = zone.input_projection(x.detach())
        spikes, _, _ = zone.neuron_groups["lif"](zin)
    wo = zone.output_projection.weight_patterns.detach().cpu().clone().requires_grad_(True)
    ref_out = O.addition_linear(spikes.cpu(), wo)
    assert torch.allclose(out.detach().cpu(), ref_out.detach(), rtol=1e-5, atol=1e-4)
    (ref_gw,) = torch.autograd.grad((ref_out * wts.cpu()).sum(), [wo])
    assert torch.allclose(got[2].cpu(), ref_gw, rtol=1e-5, atol=2e-5 * max(1.0, float(ref_gw.abs().max())))
