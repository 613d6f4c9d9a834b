"""GPU parity, round 2: the rigorous error bound of the two-stage recall (adversarial roundings,
negative strengths), write paths that name a slot twice, the shadow watermark, incremental upkeep of
the inverted lists, and the segmented centroid means."""
import pytest
import torch

from oracle import aura_oracle as O
from tests.helpers import topk_equivalent

pytestmark = pytest.mark.gpu
NOW = 1.7e9 + 777.0


def _midpoint_rows(n, D, g, up=False):
    """Unit rows whose first D-1 coordinates sit just below (above) a bf16 rounding midpoint, so that
    round-to-nearest moves every one of them by almost the full half ulp in the same direction."""
    a = 0.125 * (1.0 + 2.0 ** -8 + (2.0 ** -13 if up else -2.0 ** -13))
    sign = torch.where(torch.rand(n, D, generator=g) < 0.5, -1.0, 1.0)
    x = sign * a
    rest = 1.0 - (D - 1) * a * a
    assert rest > 0
    x[:, -1] = sign[:, -1] * rest ** 0.5
    return x.float()


def test_bound_holds_on_adversarial_roundings(dev):
    """with both operands rounded the error reaches 2^-7, twice the constant round 1 shipped.
    The kernel's per-row residual norms rho (aura_bank_shadow_update) and the query's own give a bound
    that holds on rows built to round the worst way -- and the old constant is shown to fail on them."""
    from aura_snn_rag_amd import ops
    D = 64
    g = torch.Generator().manual_seed(0)
    rows = torch.cat([_midpoint_rows(300, D, g), _midpoint_rows(300, D, g, up=True), torch.randn(9000, D, generator=g)])
    N = rows.shape[0]
    bank = rows.to(dev).contiguous()
    inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
    shadow, rho = ops.make_shadow(bank, inv)
    q = bank[:600]                                       # queries = the adversarial rows themselves
    qn = q * (1.0 / q.norm(dim=1, keepdim=True))
    qb = qn.to(torch.bfloat16).float()
    rho_q = (qb - qn).norm(dim=1) * 1.001 + (0.5 * D + 3) * 2.0 ** -24
    cos_bf16 = (qb.double() @ shadow.double().t()).float()
    cos_true = ((q.double() / q.double().norm(dim=1, keepdim=True)) @
                (bank.double() / bank.double().norm(dim=1, keepdim=True)).t()).float()
    err = (cos_bf16 - cos_true).abs()
    e_fix = 2 * D * 2.0 ** -24 + 1e-5
    bound = rho.unsqueeze(0) + rho_q.unsqueeze(1) * (1 + rho.unsqueeze(0)) + e_fix
    assert bool((err <= bound).all()), f"bound violated by {(err - bound).max().item():.3e}"
    old_constant = 2.0 ** -8 * (1 + 2.0 ** -9) + e_fix
    assert float(err.max()) > old_constant, "the case is not adversarial enough to separate the two bounds"
    assert float(rho.max()) <= 2.0 ** -8 * 1.002 + (0.5 * D + 3) * 2.0 ** -24 + 1e-7
    # end to end: the two-stage recall over this bank returns exactly what the fp32 scan returns
    meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1.0; meta[:, 1] = NOW; meta[:, 2] = -1
    qq = torch.cat([q[::7], q[::11] + 0.002 * torch.randn(q[::11].shape, generator=g).to(dev)]).contiguous()
    for k in (1, 5, 40):
        s0, i0 = ops.knn_search(bank, inv, meta, qq, k, NOW, fp32_scan=True)
        s1, i1 = ops.knn_search(bank, inv, meta, qq, k, NOW, shadow=shadow, rho=rho)
        s2, i2 = ops.knn_search(bank, inv, meta, qq, k, NOW)                      # fp32 rows, worst-case bound
        assert torch.equal(i0, i1) and torch.equal(s0, s1)
        assert torch.equal(i0, i2) and torch.equal(s0, s2)


def test_two_stage_with_negative_and_zero_strengths(dev):
    """Strengths of either sign (the metadata buffer is public): the bound stays valid because rows with
    a negative strength carry the worst-case query term; results equal the fp32 scan's."""
    from aura_snn_rag_amd import ops
    N, D, nq, k = 20000, 96, 70, 12
    g = torch.Generator().manual_seed(4)
    bank = torch.randn(N, D, generator=g).to(dev).contiguous()
    inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
    meta = torch.zeros(N, 4, device=dev)
    meta[:, 0] = (torch.rand(N, generator=g) * 2.0 - 0.6).to(dev)       # ~30 % negative
    meta[::50, 0] = 0.0
    meta[:, 1] = NOW - (3600 * torch.rand(N, generator=g)).to(dev)
    meta[:, 2] = -1
    q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.2 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
    shadow, rho = ops.make_shadow(bank, inv)
    s0, i0 = ops.knn_search(bank, inv, meta, q, k, NOW, fp32_scan=True)
    s1, i1 = ops.knn_search(bank, inv, meta, q, k, NOW, shadow=shadow, rho=rho)
    s2, i2 = ops.knn_search(bank, inv, meta, q, k, NOW)
    assert torch.equal(i0, i1) and torch.equal(s0, s1) and torch.equal(i0, i2) and torch.equal(s0, s2)
    ri, rs = O.knn_exact_batch(bank.cpu(), meta[:, 0].cpu(), meta[:, 1].cpu(), q.cpu()[:8], k, NOW)
    _, _, ok = topk_equivalent(i1[:8], s1[:8], ri, rs)
    assert ok


def _hf(M, D, **kw):
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    return HippocampalFormation(feature_dim=D, max_memories=M, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                                device="cuda", **kw)


def test_shadow_watermark_never_skips_rows(dev):
    """recall at 9000 rows, append 1000 WITHOUT a recall, then a batch that wraps the
    capacity: rows 9000..9999 must not be left unconverted behind a raised watermark."""
    D, M = 64, 12000
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(14000, D, generator=g)
    kw = dict(use_centroid_index=False, overflow="fifo")
    a, b = _hf(M, D, **kw), _hf(M, D, bf16_shadow=False, **kw)
    q = feats[torch.randint(9000, 10000, (40,), generator=g)] + 0.05 * torch.randn(40, D, generator=g)
    for hf in (a, b):
        hf.bulk_write(feats[:9000], rebuild=False)
    now = float(a.memory_metadata[0, 1].item()) + 1.0
    a.recall_batch(q, k=5, now=now)
    assert a._shadow is not None and a._shadow_valid_upto == 9000
    for hf in (a, b):
        hf.create_episodic_memories([f"p{i}" for i in range(1000)], feats[9000:10000])   # no recall in between
        hf.create_episodic_memories([f"w{i}" for i in range(2500)], feats[10000:12500])  # wraps: slots .., M-1, 0, ..
    b.memory_metadata.copy_(a.memory_metadata)
    assert a.memory_count == M and a._write_cursor == 500
    sa, ra = a.recall_batch(q, k=5, now=now)
    sb, rb = b.recall_batch(q, k=5, now=now)
    assert torch.equal(ra, rb) and torch.equal(sa, sb)
    assert a._shadow_valid_upto == M
    expect = (a.memory_features * a._inv_norm.unsqueeze(1)).to(torch.bfloat16)
    assert torch.equal(a._shadow, expect), "stale shadow rows"
    # the planted neighbours (rows 9000..9999) are found
    assert bool(((ra[:, 0] >= 9000) & (ra[:, 0] < 10000)).all())


@pytest.mark.parametrize("mode", ["reference", "fifo"])
def test_batched_overwrites_equal_sequential_writes(dev, mode):
    """a batch written into a FULL bank without the centroid index names a slot several
    times (the reference's mode: every row -> slot 0).  The last write must win, exactly as n sequential
    writes, with row, norm and metadata of the same input."""
    D, M = 32, 64
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(M + 200, D, generator=g)
    kw = dict(use_centroid_index=False, overflow=mode)
    a, b = _hf(M, D, **kw), _hf(M, D, **kw)
    ids = [f"m{i}" for i in range(M + 200)]
    a.create_episodic_memories(ids[:M], feats[:M]); b.create_episodic_memories(ids[:M], feats[:M])
    a.create_episodic_memories(ids[M:], feats[M:])                     # one batch: 200 rows into a full bank
    for i in range(M, M + 200):
        b.create_episodic_memory(ids[i], "e", feats[i])                # the reference's loop
    assert torch.equal(a.memory_features, b.memory_features)
    assert torch.equal(a._inv_norm, b._inv_norm)
    assert a.id_to_idx == b.id_to_idx and a._idx_to_id == b._idx_to_id
    if mode == "reference":
        assert torch.equal(a.memory_features[0].cpu(), feats[-1])
    assert len(a.episodic_memories) == M + 200 and a.episodic_memories["m3"].feature_idx == 3


def test_inverted_lists_follow_writes_incrementally(dev):
    """The list-sorted shadow is appended to in place after writes (and re-packed only when the slack is
    used up or the centroids are rebuilt): recall through it == a bank without shadows (fp32 lists),
    across appends, ring overwrites (holes) and a rebuild."""
    D, M = 64, 14000
    g = torch.Generator().manual_seed(12)
    centres = torch.randn(300, D, generator=g) * 3
    def draw(n):
        return centres[torch.randint(0, 300, (n,), generator=g)] + torch.randn(n, D, generator=g)
    feats = draw(10000)
    kw = dict(use_centroid_index=True, overflow="fifo")
    a, b = _hf(M, D, **kw), _hf(M, D, bf16_shadow=False, **kw)
    for hf in (a, b):
        hf.centroids_update_interval = 100000                        # no automatic rebuilds in this test
        torch.manual_seed(1)
        hf.bulk_write(feats, rebuild=True)
    b.memory_metadata[:, :2].copy_(a.memory_metadata[:, :2])
    assert torch.equal(a.memory_metadata[:, 2], b.memory_metadata[:, 2])
    now = float(a.memory_metadata[0, 1].item()) + 2.0
    q = (draw(700)).to(dev)                                          # > MASKED_SCAN_MAX_QUERIES: the lists

    def same():
        b.memory_metadata.copy_(a.memory_metadata); b.centroids.copy_(a.centroids)
        sa, ra = a.recall_batch(q, k=9, now=now)
        sb, rb = b.recall_batch(q, k=9, now=now)
        assert torch.equal(ra, rb) and torch.equal(sa, sb)
    same()
    st = a._ivf
    assert st is not None and st.valid and st.appended == 0
    packs = []
    import aura_snn_rag_amd.ops as ops
    orig = ops.bank_shadow_sorted
    def counting(*args, **kwargs):
        packs.append(1)
        return orig(*args, **kwargs)
    ops.bank_shadow_sorted = counting
    try:
        for step in range(6):                                        # appends: 6 x 40 rows, no re-pack
            new = draw(40)
            for hf in (a, b):
                hf.create_episodic_memories([f"s{step}_{i}" for i in range(40)], new)
            same()
        assert not packs and a._ivf.appended == 240 and int(a._ivf.flag.item()) == 0
        # the rows just written are found through the lists
        s, r = a.recall_batch(new[:5].to(dev).repeat(120, 1), k=1, now=now)
        assert bool((r[:5, 0] >= 10200).all())
        fill = draw(M - 10240)
        for hf in (a, b):                                            # fill the bank, then overwrite rows 0.. (holes)
            hf.create_episodic_memories([f"f{i}" for i in range(M - 10240)], fill)
        same()
        n_packs = len(packs)
        assert n_packs >= 1                                          # 3760 rows > slack: re-packed once at the recall
        for step in range(3):
            new = draw(50)
            for hf in (a, b):
                hf.create_episodic_memories([f"o{step}_{i}" for i in range(50)], new)
            same()
        assert len(packs) == n_packs and a.memory_count == M
        holes = int((a._ivf.sorted_rows[:a._ivf.n_sorted] < 0).sum()) - (a._ivf.n_sorted - M)
        listed = a._ivf.sorted_rows[a._ivf.sorted_rows >= 0]
        assert listed.numel() == M and torch.equal(torch.sort(listed).values, torch.arange(M, device=dev, dtype=torch.int32))
        assert holes >= 0
        torch.manual_seed(5); a.rebuild_centroids(perm=torch.randperm(M, generator=torch.Generator().manual_seed(3)))
        b.rebuild_centroids(perm=torch.randperm(M, generator=torch.Generator().manual_seed(3)))
        same()
        assert len(packs) == n_packs + 1
        # one list outgrows its slack (700 near-copies of one row, slack 512): the append kernel drops the
        # overflowing rows and raises the lists' flag, the next recall sees it in its one flag read, re-packs
        # and repeats -- no result ever comes from stale lists
        burst = a.memory_features[123].cpu() + 0.01 * torch.randn(700, D, generator=g)
        for j in range(0, 700, 100):
            for hf in (a, b):
                hf.create_episodic_memories([f"b{j + i}" for i in range(100)], burst[j:j + 100])
        assert a._ivf.valid and int(a._ivf.flag.item()) == 1
        same()
        assert len(packs) == n_packs + 2 and int(a._ivf.flag.item()) == 0
    finally:
        ops.bank_shadow_sorted = orig


def test_segment_means_match_masked_means(dev):
    """rebuild_centroids' means as a segmented reduction: with the assignment given, the means equal the
    reference's masked means (hippocampal.py:358-363) to fp32 summation-order accuracy, counts and
    metadata ids exactly; empty clusters keep their centroid; reproducible bit for bit."""
    from aura_snn_rag_amd import ops
    for N, D, k in ((5000, 64, 40), (70001, 768, 256), (300, 8, 256), (4097, 100, 7)):
        g = torch.Generator().manual_seed(N)
        bank = (torch.randn(N, D, generator=g) * 3 + 1).to(dev).contiguous()
        assign = torch.randint(0, k, (N,), generator=g).to(torch.int32)
        assign[assign == 3] = 4                                        # cluster 3 stays empty
        assign = assign.to(dev)
        cent = torch.full((256, D), 7.0, device=dev)
        meta = torch.zeros(N, 4, device=dev)
        counts = torch.zeros(256, device=dev)
        order, seg_off = ops.kmeans_update(bank, assign, cent, k, counts=counts, meta=meta, update_means=True)
        ref = torch.full((256, D), 7.0, dtype=torch.float64)
        bc = bank.cpu().double(); ac = assign.cpu().long()
        for c in range(k):
            m = ac == c
            if m.any():
                ref[c] = bc[m].mean(dim=0)
        assert torch.allclose(cent.cpu().double(), ref, rtol=1e-5, atol=1e-5)
        assert torch.equal(cent[3], torch.full((D,), 7.0, device=dev))
        assert torch.equal(counts[:k].cpu(), torch.bincount(ac, minlength=k)[:k].float())
        assert torch.equal(meta[:, 2].cpu(), ac.float())
        cent2 = torch.full((256, D), 7.0, device=dev)
        ops.kmeans_update(bank, assign, cent2, k, update_means=True)
        assert torch.equal(cent, cent2)


def test_rebuild_means_exact_given_the_assignment(dev):
    """rebuild_centroids' centroids compared at tight tolerance: with the GPU's own first
    assignment the reference's masked-mean arithmetic (oracle) must reproduce the centroids."""
    from aura_snn_rag_amd import ops
    hf = _hf(6000, 64)
    g = torch.Generator().manual_seed(3)
    centers = torch.randn(40, 64, generator=g) * 3
    feats = centers[torch.randint(0, 40, (5000,), generator=g)] + torch.randn(5000, 64, generator=g)
    hf.use_centroid_index = False
    hf.create_episodic_memories([f"m{i}" for i in range(5000)], feats)
    hf.use_centroid_index = True
    perm = torch.randperm(5000, generator=g)
    hf.rebuild_centroids(perm=perm)
    # replay with the kernel's assignment: init -> assign (GPU) -> masked means (CPU, reference arithmetic)
    cent0 = torch.zeros(256, 64); cent0[:256] = feats[perm[:256]]
    assign1 = ops.kmeans_assign(hf.memory_features, cent0.to(dev), 5000, 256).cpu().long()
    ref = cent0.clone()
    for c in range(256):
        m = assign1 == c
        if m.any():
            ref[c] = feats[m].mean(dim=0)
    assert torch.allclose(hf.centroids.cpu(), ref, rtol=1e-5, atol=1e-5)
    # and the oracle's own assignment agrees with the kernel's almost everywhere
    d = torch.cdist(feats, cent0)
    agree = (torch.argmin(d, dim=1) == assign1).float().mean().item()
    assert agree >= 0.995, f"assignment agreement {agree}"
    assign2 = hf.memory_metadata[:5000, 2].cpu().long()
    assert torch.equal(hf.centroid_counts.cpu(), torch.bincount(assign2, minlength=256).float())
