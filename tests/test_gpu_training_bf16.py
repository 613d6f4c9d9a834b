"""bf16 forms of the prosody GIF and of the LIF step (VERDICT r02 "missing" #6): forward bit-exact against the
oracle's restatement run on bf16 CPU tensors (torch rounds every op to bf16 there, as the reference's bf16
tensors do), mixed fp32/bf16 calls promoted as the reference's eager ops promote them, and the surrogate
gradient kernels against (1) the intended arithmetic -- fp32 BPTT from the saved bf16 values with the forward's
roundings re-applied -- to one bf16 rounding and (2) the reference's own bf16 autograd graph within bf16 noise."""
import math

import pytest
import torch

from oracle import aura_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rb(x):
    return x.to(BF).to(torch.float32)


def _f(x):
    return x.to(torch.float32)


def _one_rounding(got, want, name, frac_ok=0.999):
    a32 = got.float().cpu()
    err = (a32 - want).abs()
    tol = 2.0 ** -7 * want.abs() + 1e-6
    frac = float((err <= tol).float().mean())
    assert frac > frac_ok, f"{name}: {1 - frac:.2e} of the entries beyond one bf16 rounding (max err {err.max():.3e})"


def _bf16_noise(got, ref, name, cos_min=0.995, rel_max=0.08):
    a32, b32 = got.float().cpu().flatten(), ref.float().flatten()
    if float(b32.norm()) == 0.0:
        assert float(a32.norm()) == 0.0, name
        return
    cos = float(torch.dot(a32, b32) / (a32.norm() * b32.norm()))
    rel = float((a32 - b32).norm() / b32.norm())
    assert cos > cos_min and rel < rel_max, f"{name}: cosine {cos:.5f}, relative L2 distance {rel:.4f} to the bf16 autograd"


# ---- prosody GIF ----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("B,T,I,H", [(4, 10, 24, 48), (3, 7, 9, 21), (16, 32, 64, 256)])
def test_prosody_gif_bf16_forward_is_the_per_op_rounded_loop(dev, B, T, I, H):
    from aura_snn_rag_amd.core.language_zone.prosody_gif import ProsodyModulatedGIF
    torch.manual_seed(B + T)
    pg = ProsodyModulatedGIF(I, H, L=8, alpha=0.05, attention_modulation_strength=0.3).to(dev).to(BF)
    x = (torch.randn(B, T, I, device=dev) * 3).to(BF)
    gains = (0.2 + 3.0 * torch.rand(B, T)).to(BF).to(dev)
    thr = pg.threshold
    with torch.no_grad():
        h = pg.linear(x).cpu()
        assert h.dtype == BF
        for gn in (gains, None):
            s, (v, th) = pg(x, attention_gains=gn)
            assert s.dtype == BF and v.dtype == BF and th.dtype == BF
            rs, rv, rt = O.prosody_gif_run(h, torch.zeros(B, H, dtype=BF), torch.full((B, H), thr, dtype=BF),
                                           None if gn is None else gn.cpu(), pg.decay, 8, 0.05, thr, 0.3)
            assert rs.dtype == BF
            assert torch.equal(s.cpu(), rs) and torch.equal(v.cpu(), rv) and torch.equal(th.cpu(), rt)
            s2, (v2, t2) = pg(x, state=(v, th), attention_gains=gn)
            rs2, rv2, rt2 = O.prosody_gif_run(h, rv, rt, None if gn is None else gn.cpu(), pg.decay, 8, 0.05, thr, 0.3)
            assert torch.equal(s2.cpu(), rs2) and torch.equal(v2.cpu(), rv2) and torch.equal(t2.cpu(), rt2)
        # the recording forward is the same loop
    xg = x.clone().requires_grad_(True)
    s3, (v3, t3) = pg(xg, attention_gains=gains)
    assert s3.requires_grad
    with torch.no_grad():
        s4, (v4, t4) = pg(x, attention_gains=gains)
    assert torch.equal(s3.detach(), s4) and torch.equal(v3.detach(), v4) and torch.equal(t3.detach(), t4)


def test_prosody_gif_mixed_dtypes_promote_like_the_reference(dev):
    """fp32 gains with a bf16 module, and a bf16 current from autocast with the fp32 state the reference
    creates from ``x.dtype``: the reference's eager ops promote the whole loop to fp32."""
    from aura_snn_rag_amd.core.language_zone.prosody_gif import ProsodyModulatedGIF
    torch.manual_seed(5)
    B, T, I, H = 6, 9, 16, 40
    pg = ProsodyModulatedGIF(I, H, L=8, alpha=0.05, attention_modulation_strength=0.3).to(dev).to(BF)
    x = (torch.randn(B, T, I, device=dev) * 3).to(BF)
    gains32 = (0.2 + 3.0 * torch.rand(B, T)).to(dev)
    with torch.no_grad():
        h = pg.linear(x).cpu()
        s, (v, th) = pg(x, attention_gains=gains32)
        assert s.dtype == torch.float32 and v.dtype == torch.float32
        rs, rv, rt = O.prosody_gif_run(h, torch.zeros(B, H, dtype=BF), torch.full((B, H), pg.threshold, dtype=BF),
                                       gains32.cpu(), pg.decay, 8, 0.05, pg.threshold, 0.3)
        assert rs.dtype == torch.float32
        assert torch.equal(s.cpu(), rs) and torch.equal(v.cpu(), rv) and torch.equal(th.cpu(), rt)
    pg32 = ProsodyModulatedGIF(I, H, L=8, alpha=0.05, attention_modulation_strength=0.3).to(dev)
    x32 = torch.randn(B, T, I, device=dev) * 3
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        h = pg32.linear(x32)
        assert h.dtype == BF
        s, (v, th) = pg32(x32, attention_gains=gains32)
    rs, rv, rt = O.prosody_gif_run(h.cpu(), torch.zeros(B, H), torch.full((B, H), pg32.threshold), gains32.cpu(),
                                   pg32.decay, 8, 0.05, pg32.threshold, 0.3)
    assert s.dtype == torch.float32
    assert torch.equal(s.cpu(), rs) and torch.equal(v.cpu(), rv) and torch.equal(th.cpu(), rt)


def _prosody_bf16_bptt_fp32_math(h, gains, v0, t0, ws, wv, wt, decay, L, alpha, thr0, strength):
    """What aura_gif_prosody_backward_bf16 is meant to compute, in torch on the CPU."""
    rows, T, H = h.shape
    mod = gains is not None
    v, th = _f(v0), _f(t0)
    A, TH = [], []

    def step_consts(t):
        if not mod:
            return 1.0, 1.0, alpha, False
        g = _f(gains[:, t]).unsqueeze(1)
        raw = _rb(1.0 - _rb(strength * _rb(g - 1.0)))
        return g, torch.clamp(raw, 0.5, 1.5), _rb(alpha * g), (raw >= 0.5) & (raw <= 1.5)

    def mid(a, thp, scale):
        te = _rb(thp * scale) if mod else thp
        cl = _rb(_rb(L * te) * 2.0)
        b = torch.minimum(torch.maximum(a, -cl), cl)
        n = _rb(b / te)
        s = torch.clamp(torch.floor(n), 0, L)
        return te, cl, b, n, s

    for t in range(T):
        g, scale, ae, _ = step_consts(t)
        i_t = _rb(_f(h[:, t]) * g) if mod else _f(h[:, t])
        a = _rb(_rb(v * decay) + i_t)
        te, cl, b, n, s = mid(a, th, scale)
        A.append(a); TH.append(th)
        v = _rb(b - _rb(s * te))
        if alpha > 0:
            th = _rb(_rb(th + _rb(ae * s)) - _rb(ae * _rb(th - thr0)))
    gv, gth = _f(wv).clone(), _f(wt).clone()
    gh = torch.zeros(rows, T, H)
    gg = torch.zeros(rows, T) if mod else None
    for t in range(T - 1, -1, -1):
        a, thp = A[t], TH[t]
        g, scale, ae, live = step_consts(t)
        te, cl, b, n, s = mid(a, thp, scale)
        gs = _f(ws[:, t]).clone()
        gthp = gth.clone()
        acc = torch.zeros(rows, H)
        if alpha > 0:
            gs = gs + ae * gth
            gthp = gthp - ae * gth
            if mod:
                acc = acc + gth * (s - (thp - thr0)) * alpha
        gs = gs - te * gv
        gte = -s * gv
        gb = gv.clone()
        tri = torch.clamp(1.0 - 2.0 * (n - torch.round(n)).abs(), 0.0, 1.0)
        sur = torch.where((n >= 0) & (n <= L + 1.0), tri, torch.zeros_like(tri))
        gn = gs * sur
        gb = gb + gn / te
        gte = gte + (-gn * b / (te * te))
        lo, hi = a < -cl, a > cl
        ga = torch.where(lo | hi, torch.zeros_like(gb), gb)
        gcl = torch.where(lo, -gb, torch.where(hi, gb, torch.zeros_like(gb)))
        gte = gte + gcl * (2.0 * L)
        gthp = gthp + (gte * scale if mod else gte)
        if mod:
            acc = acc + torch.where(live, gte * thp * (-strength), torch.zeros_like(gte))
            acc = acc + ga * _f(h[:, t])
            gg[:, t] = acc.sum(dim=1)
        gh[:, t] = ga * g if mod else ga
        gv = ga * decay
        gth = gthp
    return gh, gg, gv, gth


@pytest.mark.parametrize("rows,T,H,L,alpha,use_gains", [(7, 12, 64, 8, 0.05, True), (3, 5, 37, 4, 0.05, True),
                                                        (16, 16, 128, 16, 0.02, True), (8, 10, 64, 8, 0.05, False),
                                                        (8, 10, 64, 8, 0.0, True)])
def test_prosody_gif_bptt_bf16(dev, rows, T, H, L, alpha, use_gains):
    from aura_snn_rag_amd.core.language_zone.prosody_gif import ProsodyGifFunction
    g = torch.Generator().manual_seed(rows * 31 + T * 5 + H)
    decay, thr0, strength = math.exp(-0.1), 1.0, 0.3
    h = (torch.randn(rows, T, H, generator=g) * 3).to(BF)
    gains = (0.2 + 3.0 * torch.rand(rows, T, generator=g)).to(BF) if use_gains else None
    v0 = (0.4 * torch.randn(rows, H, generator=g)).to(BF)
    t0 = (1.0 + 0.3 * torch.rand(rows, H, generator=g)).to(BF)
    ws, wv, wt = (torch.randn(s, generator=g).to(BF) for s in ((rows, T, H), (rows, H), (rows, H)))
    leaves = [x.to(dev).requires_grad_(True) for x in (h, v0, t0)] + ([gains.to(dev).requires_grad_(True)] if use_gains else [])
    sd, vT, tT = ProsodyGifFunction.apply(leaves[0], leaves[3] if use_gains else None, leaves[1], leaves[2], decay, L,
                                          alpha, thr0, strength)
    assert sd.dtype == BF and vT.dtype == BF
    got = torch.autograd.grad((sd * ws.to(dev)).sum() + (vT * wv.to(dev)).sum() + (tT * wt.to(dev)).sum(), leaves)
    assert all(x.dtype == BF for x in got)
    # the reference's bf16 graph: same spikes / state ...
    cpu = [x.clone().requires_grad_(True) for x in (h, v0, t0)] + ([gains.clone().requires_grad_(True)] if use_gains else [])
    s, v, th = O.prosody_gif_run_grad(cpu[0], cpu[1], cpu[2], cpu[3] if use_gains else None, decay, L, alpha, thr0, strength)
    assert torch.equal(sd.detach().cpu(), s.detach()) and torch.equal(vT.detach().cpu(), v.detach())
    assert torch.equal(tT.detach().cpu(), th.detach())
    # (1) the intended arithmetic, to one bf16 rounding of the result
    gh, gg, gv, gth = _prosody_bf16_bptt_fp32_math(h, gains, v0, t0, ws, wv, wt, decay, L, alpha, thr0, strength)
    _one_rounding(got[0], gh, "g_h")
    _one_rounding(got[1], gv, "g_v0")
    _one_rounding(got[2], gth, "g_theta0")
    if use_gains:
        # a sum over H channels of mixed sign: the rounding is relative to the summed magnitude, not to the result
        err = (got[3].float().cpu() - gg).abs()
        assert float((err <= 2.0 ** -7 * gg.abs() + 1e-4 * float(gg.abs().max())).float().mean()) > 0.999, float(err.max())
    # (2) ... and gradients within bf16 noise of its autograd
    ref = torch.autograd.grad((s * ws).sum() + (v * wv).sum() + (th * wt).sum(), cpu)
    for a, b, name in zip(got, ref, ("g_h", "g_v0", "g_theta0", "g_gains")):
        _bf16_noise(a, b, name, cos_min=0.99, rel_max=0.15)


def test_prosody_gif_module_trains_in_bf16(dev):
    from aura_snn_rag_amd.core.language_zone.prosody_gif import ProsodyModulatedGIF
    torch.manual_seed(0)
    pg = ProsodyModulatedGIF(32, 64, L=8, alpha=0.05).to(dev).to(BF)
    x = torch.randn(4, 6, 32, device=dev, dtype=BF, requires_grad=True)
    gains = (0.5 + torch.rand(4, 6, device=dev)).to(BF).requires_grad_(True)
    s, (v, th) = pg(x, attention_gains=gains)
    (s.float().mean() + v.float().mean() + th.float().mean()).backward()
    for t in (x.grad, gains.grad, pg.linear.weight.grad, pg.linear.bias.grad):
        assert t is not None and t.dtype == BF and bool(torch.isfinite(t.float()).all())
    assert float(pg.linear.weight.grad.float().abs().sum()) > 0 and float(gains.grad.float().abs().sum()) > 0
    # fp32 gains with the bf16 module: promoted loop, gradients come back in each leaf's own dtype
    x2 = torch.randn(4, 6, 32, device=dev, dtype=BF, requires_grad=True)
    g2 = (0.5 + torch.rand(4, 6, device=dev)).requires_grad_(True)
    s, (v, th) = pg(x2, attention_gains=g2)
    assert s.dtype == torch.float32
    (s.mean() + v.mean()).backward()
    assert x2.grad.dtype == BF and g2.grad.dtype == torch.float32


# ---- LIF ------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape,size", [((5, 48), 48), ((2, 3, 33), 33), ((1, 4), 4), ((64, 1024), 1024)])
def test_lif_bf16_forward_and_gradients(dev, shape, size):
    from aura_snn_rag_amd.base.neuron import VectorizedLIFNeuron
    g = torch.Generator().manual_seed(size)
    lif = VectorizedLIFNeuron(size, beta=0.9, threshold=0.6, init_slope=4.0).to(dev).to(BF)
    with torch.no_grad():
        lif.slope.copy_((2 + 6 * torch.rand(size, generator=g)).to(BF).to(dev))
    assert lif.beta.dtype == BF and lif.slope.dtype == BF
    xs = [torch.randn(shape, generator=g).to(BF) for _ in range(3)]
    ws = [torch.randn(shape, generator=g).to(BF) for _ in range(4)]
    beta, thr = lif.beta.cpu(), lif.threshold.cpu()
    # inference: three single steps, then the same as one sequence launch
    mem = torch.zeros(shape, dtype=BF)
    ref_spk = []
    with torch.no_grad():
        for t in range(3):
            spk, m = lif(xs[t].to(dev))
            rs, mem = O.lif_step(xs[t], mem, beta, thr)
            assert spk.dtype == BF and torch.equal(spk.cpu(), rs) and torch.equal(m.cpu(), mem)
            ref_spk.append(rs)
        if len(shape) == 2:
            lif.reset_mem()
            seq = lif.forward_sequence(torch.stack(xs, dim=1).to(dev))
            assert torch.equal(seq.cpu(), torch.stack(ref_spk, dim=1)) and torch.equal(lif.mem.cpu(), mem)
    # training: the reference's bf16 autograd graph over three steps
    xo = [x.clone().requires_grad_(True) for x in xs]
    slope = lif.slope.detach().cpu().clone().requires_grad_(True)
    mem, loss = torch.zeros(shape, dtype=BF), 0.0
    for t in range(3):
        spk, mem = O.lif_step_grad(xo[t], mem, beta, thr, slope)
        loss = loss + (spk * ws[t]).sum()
    loss = loss + (mem * ws[3]).sum()
    ref = torch.autograd.grad(loss, xo + [slope])
    lif.reset_mem()
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    loss = 0.0
    for t in range(3):
        spk, m = lif(xd[t])
        assert spk.dtype == BF and torch.equal(spk.detach().cpu(), ref_spk[t])
        loss = loss + (spk * ws[t].to(dev)).sum()
    assert torch.equal(m.detach().cpu(), mem.detach())
    loss = loss + (m * ws[3].to(dev)).sum()
    got = torch.autograd.grad(loss, xd + [lif.slope])
    assert all(a.dtype == BF for a in got)
    for a, b, name in zip(got, ref, ("g_x0", "g_x1", "g_x2", "g_slope")):
        _bf16_noise(a, b, name, cos_min=0.99, rel_max=0.15)


def test_lif_bf16_backward_is_the_fp32_formula_rounded_once(dev):
    from aura_snn_rag_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C = 37, 200
    pre = torch.randn(B, C, generator=g).to(BF)
    gs, gm = torch.randn(B, C, generator=g).to(BF), torch.randn(B, C, generator=g).to(BF)
    beta = (0.5 + 0.5 * torch.rand(C, generator=g)).to(BF)
    thr = (0.3 + torch.rand(C, generator=g)).to(BF)
    slope = (2 + 6 * torch.rand(C, generator=g)).to(BF)
    d = lambda t: t.to(dev)
    g_x, g_prev = torch.empty(B, C, dtype=BF, device=dev), torch.empty(B, C, dtype=BF, device=dev)
    raw = torch.empty(B, C, device=dev)
    ops.lif_backward(d(pre), d(gs), d(gm), d(beta), d(thr), d(slope), g_x, g_prev, raw)
    p, s_, m_, bt, th, sl = (_f(t) for t in (pre, gs, gm, beta, thr, slope))
    g_s = s_ - m_ * th
    den = (sl * p).abs() + 1.0
    g_m = m_ + g_s * (sl / (den * den))
    den2 = sl * p.abs() + 1.0
    want_raw = -g_s * p.abs() * torch.sign(p) / (den2 * den2)
    _one_rounding(g_x, g_m, "g_x")
    _one_rounding(g_prev, bt * g_m, "g_mem_prev")
    assert torch.allclose(raw.cpu(), want_raw, rtol=1e-5, atol=1e-6)


def test_lif_mixed_dtypes_promote_like_the_reference(dev):
    """bf16 input into an fp32 module (autocast upstream) and fp32 input into a bf16 module: ``beta * mem + input``
    promotes to fp32 in the reference; the result is fp32 and bit-equal to the oracle run with the same mix."""
    from aura_snn_rag_amd.base.neuron import VectorizedLIFNeuron
    g = torch.Generator().manual_seed(11)
    size = 96
    xs = [torch.randn(7, size, generator=g) for _ in range(3)]
    for mod_dt, x_dt in ((torch.float32, BF), (BF, torch.float32)):
        lif = VectorizedLIFNeuron(size, beta=0.9, threshold=0.6).to(dev).to(mod_dt)
        beta, thr = lif.beta.cpu(), lif.threshold.cpu()
        mem = None
        with torch.no_grad():
            for t in range(3):
                x = xs[t].to(x_dt)
                if mem is None:
                    mem = torch.zeros_like(x)
                spk, m = lif(x.to(dev))
                rs, mem = O.lif_step(x, mem, beta, thr)
                assert rs.dtype == spk.dtype == torch.float32 and mem.dtype == m.dtype == torch.float32
                assert torch.equal(spk.cpu(), rs) and torch.equal(m.cpu(), mem)
