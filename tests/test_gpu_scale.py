"""GPU parity where the metric lives: a 1M x 768 bank on one GPU (exact two-stage recall, centroid
index through probe masks / inverted lists on the two-stage scan / fp32 lists) and the config-4
shard shape (125 000 x 768, 2048 queries, idx_base != 0), against the CPU oracle.  Every test prints
how many queries are index-exact (run with -s to see it; the counts are asserted as well)."""
import pytest
import torch

from oracle import aura_oracle as O
from tests.helpers import record_parity, topk_equivalent

pytestmark = pytest.mark.gpu


def _fill(hf, rows, D, seed, dev, strengths=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    for r0 in range(0, rows, 1 << 17):
        n = min(1 << 17, rows - r0)
        hf.bulk_write(torch.randn(n, D, generator=g, device=dev), rebuild=False)
    if strengths:                                       # decayed bank: ranking != cosine ranking
        hf.memory_metadata[:rows, 0] = 0.5 + 0.5 * torch.rand(rows, generator=g, device=dev)


@pytest.fixture(scope="module")
def bank_1m(dev):
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    N, D = 1_000_000, 768
    hf = HippocampalFormation(feature_dim=D, max_memories=N, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                              device="cuda", use_centroid_index=True)
    _fill(hf, N, D, 1234, dev, strengths=True)
    hf.rebuild_centroids(perm=torch.randperm(N, generator=torch.Generator().manual_seed(7)))
    now = float(hf.memory_metadata[0, 1].item()) + 30.0
    g = torch.Generator(device=dev).manual_seed(5)
    pick = torch.randint(0, N, (150,), generator=g, device=dev)
    q = torch.cat([hf.memory_features[pick] + 0.05 * torch.randn(150, D, generator=g, device=dev),
                   torch.randn(150, D, generator=g, device=dev)]).contiguous()
    host = dict(bank=hf.memory_features.cpu(), meta=hf.memory_metadata.cpu(), cent=hf.centroids.cpu())
    yield hf, q, pick, now, host
    del hf
    torch.cuda.empty_cache()


def test_1m_exact_recall_vs_oracle(bank_1m, dev):
    """Exact recall at 1M x 768: two-stage (bf16 shadow) == all-fp32 scan bit for bit on 300 queries; the
    oracle on 12 of them."""
    from aura_snn_rag_amd import ops
    hf, q, pick, now, host = bank_1m
    N, k = hf.memory_count, 32
    s2, r2 = hf.recall_batch(q, k=k, now=now, use_candidates=False)
    assert hf._shadow is not None
    s0, r0 = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N, fp32_scan=True)
    assert torch.equal(r0, r2) and torch.equal(s0, s2)
    assert bool((r2[:150, 0] == pick.to(torch.int32)).all())
    sub = torch.cat([torch.arange(0, 32), torch.arange(268, 300)])      # 32 planted + 32 random queries
    ri, rs = O.knn_exact_batch(host["bank"], host["meta"][:, 0], host["meta"][:, 1], q[sub].cpu(), k, now)
    exact, n, ok = topk_equivalent(r2[sub], s2[sub], ri, rs)
    print(f"\n[1M x 768 exact recall] index-exact queries vs oracle: {exact}/{n}")
    # ok: every row that differs from the oracle's is one whose oracle score ties the oracle's pick within 2e-6
    record_parity("1Mx768_exact_recall_k32", exact, n, score_near_ties=n - exact if ok else 0)
    assert ok, "a row differs from the oracle's beyond an fp32 near-tie"
    sc = s2.cpu()
    assert bool((sc[:, :-1] >= sc[:, 1:]).all()) and bool((r2 >= 0).all()) and bool((r2 < N).all())


def test_1m_centroid_index_paths_vs_oracle(bank_1m, dev):
    """Centroid-index recall at 1M x 768: inverted lists on the two-stage scan (the product's path) ==
    fp32 inverted lists == masked two-stage scan == masked fp32 scan, bit for bit on 300 queries; the
    oracle's candidate path on 64 of them (queries whose 8th / 9th nearest centroids are an fp32
    near-tie are compared against the masked fp32 scan only: the probe set itself is ambiguous)."""
    from aura_snn_rag_amd import ops
    hf, q, pick, now, host = bank_1m
    N, k = hf.memory_count, 32
    s_p, r_p = hf.recall_batch(q, k=k, now=now)                                  # product: lists on the two-stage scan
    assert hf._ivf is not None and hf._ivf.valid
    s_m, r_m = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N,
                              centroids=hf.centroids, nprobe=8, fp32_scan=True)
    assert torch.equal(r_p, r_m) and torch.equal(s_p, s_m)
    shadow = hf._ensure_shadow()
    s_t, r_t = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N,
                              centroids=hf.centroids, nprobe=8, shadow=shadow, rho=hf._rho)
    assert torch.equal(r_t, r_m) and torch.equal(s_t, s_m)
    list_rows, list_off, list_len, longest = hf._ensure_lists()
    cap = ops.ivf_capacity(longest, k)
    if cap is not None:
        s_l, r_l, ovf = ops.knn_search_ivf(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, N,
                                           hf.centroids, 8, list_rows, list_off, list_len, cap)
        assert int(ovf.item()) == 0 and torch.equal(r_l, r_m) and torch.equal(s_l, s_m)
    ob = O.OracleBank(1, 768)
    ob.M = N; ob.features = host["bank"]; ob.metadata = host["meta"]; ob.centroids = host["cent"]
    ob.index_ready = True; ob.count = N
    sub = list(range(0, 32)) + list(range(268, 300))                  # 32 planted + 32 random queries
    exact = ties = 0
    for j in sub:
        qq = q[j].cpu()
        d = torch.sort(torch.norm(ob.centroids - qq, dim=1)).values
        if float(d[8] - d[7]) <= 2e-6 * float(d[7]):
            ties += 1
            continue
        rows, sc = ob.recall(qq, k, now)
        e, n, ok = topk_equivalent(r_p[j:j + 1], s_p[j:j + 1], rows.unsqueeze(0), sc.unsqueeze(0))
        assert ok, f"query {j}"
        exact += e
    print(f"\n[1M x 768 centroid-index recall] index-exact queries vs oracle: {exact}/{len(sub) - ties} "
          f"({ties} probe near-ties skipped)")
    # (every compared query passed `ok` above: rows equal except at score near-ties of the oracle's own scores)
    record_parity("1Mx768_centroid_index_recall_k32", exact, len(sub) - ties, score_near_ties=len(sub) - ties - exact,
                  probe_near_ties=ties)
    assert len(sub) - ties >= 48, "too many probe near-ties to call this a parity check"


def test_config4_shard_shape_vs_oracle(dev):
    """One rank's share of BASELINE config 4: 125 000 x 768 rows, the 8 x 256 = 2048 all-gathered queries,
    top-32, rows reported with idx_base = first global row of the shard."""
    from aura_snn_rag_amd import ops
    N, D, nq, k, base = 125_000, 768, 2048, 32, 3 * 125_000
    g = torch.Generator().manual_seed(44)
    bank = torch.randn(N, D, generator=g)
    meta = torch.zeros(N, 4); meta[:, 0] = 0.5 + 0.5 * torch.rand(N, generator=g); meta[:, 1] = 1.7e9; meta[:, 2] = -1
    pick = torch.randint(0, N, (nq // 2,), generator=g)
    q = torch.cat([bank[pick] + 0.05 * torch.randn(nq // 2, D, generator=g), torch.randn(nq - nq // 2, D, generator=g)])
    b = bank.to(dev).contiguous(); m = meta.to(dev).contiguous(); qd = q.to(dev).contiguous()
    inv = torch.empty(N, device=dev); ops.bank_row_norms(b, inv, 0, N)
    shadow, rho = ops.make_shadow(b, inv)
    s1, i1 = ops.knn_search(b, inv, m, qd, k, 1.7e9, idx_base=base, shadow=shadow, rho=rho)
    s0, i0 = ops.knn_search(b, inv, m, qd, k, 1.7e9, idx_base=base, fp32_scan=True)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    assert bool((i1 >= base).all()) and bool((i1 < base + N).all())
    assert bool((i1[: nq // 2, 0].cpu() == (pick + base).to(torch.int32)).all())
    sub = torch.cat([torch.arange(0, 32), torch.arange(nq - 32, nq)])
    ri, rs = O.knn_exact_batch(bank, meta[:, 0], meta[:, 1], q[sub], k, 1.7e9)
    exact, n, ok = topk_equivalent(i1[sub] - base, s1[sub], ri, rs)
    print(f"\n[config-4 shard 125000 x 768, 2048 queries, idx_base {base}] index-exact queries vs oracle: {exact}/{n}")
    record_parity("config4_shard_125000x768_k32", exact, n, score_near_ties=n - exact if ok else 0)
    assert ok


# ----------------------------------------------------------------------------------------------
# The row-sharded bank on real kernels: two processes share the one GPU of the test box and talk
# through gloo (RCCL refuses two ranks on one device); every rank drives the HIP library.
# ----------------------------------------------------------------------------------------------
def _sharded_rows(world):
    # every shard must hold at least 8192 rows for the two-stage (and hence the staged, bound-exchanging) recall
    return 20000 if world == 2 else 12000 * world - 2000


def _sharded_gpu_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    from aura_snn_rag_amd.sharded import ShardedHippocampus
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    D, M, NF = 64, 12000 * world, _sharded_rows(world)
    g = torch.Generator().manual_seed(31)
    centres = torch.randn(300, D, generator=g) * 3
    feats = centres[torch.randint(0, 300, (NF,), generator=g)] + torch.randn(NF, D, generator=g)
    extra = centres[torch.randint(0, 300, (300,), generator=g)] + torch.randn(300, D, generator=g)
    q = centres[torch.randint(0, 300, (700,), generator=g)] + torch.randn(700, D, generator=g)
    perm = torch.randperm(NF, generator=g)
    local = HippocampalFormation(feature_dim=D, max_memories=M // world, n_place_cells=8, n_time_cells=4,
                                 n_grid_cells=4, device="cuda", use_centroid_index=True, overflow="fifo")
    local.centroids_update_interval = 10 ** 9
    now = [1.7e9]
    sh = ShardedHippocampus(local, M, now_fn=lambda: now[0])
    for i in range(0, NF, 7000):                                      # batches straddle the shard boundaries (12000 rows each)
        sh.write([f"m{j}" for j in range(i, min(i + 7000, NF))], feats[i:i + 7000])
    sh.rebuild_centroids(perm=perm)
    sh.write([f"x{j}" for j in range(300)], extra)                    # online centroid updates on the replicas
    s_c, r_c = sh.recall_batch(q.to(dev), k=9, now=now[0])            # > 512 queries: inverted lists per shard
    s_e, r_e = sh.recall_batch(q.to(dev), k=9, now=now[0], use_candidates=False)
    per = 700 // world
    myq = q[rank * per:(rank + 1) * per].to(dev).contiguous()
    s_o, r_o = sh.recall_batch(myq, k=9, now=now[0], all_gather_queries=True)
    torch.save(dict(count=sh.memory_count, local=local.memory_count, cent=local.centroids.cpu(), exchanges=sh.exchanges,
                    res=[t.cpu() for t in (s_c, r_c, s_e, r_e)], own=(s_o.cpu(), r_o.cpu()),
                    meta=local.memory_metadata.cpu()), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])      # 4 shards: the bound exchanges run with k2 = ceil(9 / 4) = 3
def test_sharded_bank_on_hip_kernels_two_ranks_one_gpu(dev, tmp_path, world):
    import socket
    import torch.multiprocessing as mp
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    out = str(tmp_path / "shgpu")
    mp.get_context("spawn")
    mp.spawn(_sharded_gpu_worker, args=(world, port, out), nprocs=world, join=True)
    parts = [torch.load(out + f".{r}") for r in range(world)]
    # the same operations on ONE bank
    D, M, NF = 64, 12000 * world, _sharded_rows(world)
    g = torch.Generator().manual_seed(31)
    centres = torch.randn(300, D, generator=g) * 3
    feats = centres[torch.randint(0, 300, (NF,), generator=g)] + torch.randn(NF, D, generator=g)
    extra = centres[torch.randint(0, 300, (300,), generator=g)] + torch.randn(300, D, generator=g)
    q = centres[torch.randint(0, 300, (700,), generator=g)] + torch.randn(700, D, generator=g)
    perm = torch.randperm(NF, generator=g)
    from aura_snn_rag_amd.core import hippocampal as H
    hf = HippocampalFormation(feature_dim=D, max_memories=M, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                              device="cuda", use_centroid_index=True, overflow="fifo")
    hf.centroids_update_interval = 10 ** 9
    hf.create_episodic_memories([f"m{j}" for j in range(NF)], feats)
    hf.rebuild_centroids(perm=perm)
    hf.create_episodic_memories([f"x{j}" for j in range(300)], extra)
    hf.memory_metadata[:, 1] = 1.7e9                                  # the sharded run's clock
    s_c, r_c = hf.recall_batch(q, k=9, now=1.7e9)
    s_e, r_e = hf.recall_batch(q, k=9, now=1.7e9, use_candidates=False)
    R, total = M // world, NF + 300
    assert parts[0]["count"] == total and [p["local"] for p in parts] == [max(0, min(R, total - r * R)) for r in range(world)]
    assert all(p["exchanges"] >= 2 for p in parts), "the shards' bounds were not exchanged"
    assert all(torch.equal(parts[0]["cent"], p["cent"]) for p in parts[1:])
    assert torch.allclose(parts[0]["cent"], hf.centroids.cpu(), rtol=1e-5, atol=1e-5)
    meta = torch.cat([p["meta"] for p in parts])
    agree = (meta[:total, 2] == hf.memory_metadata[:total, 2].cpu()).float().mean().item()
    assert agree >= 0.999, f"centroid ids: {agree}"
    ps_c, pr_c, ps_e, pr_e = parts[0]["res"]
    assert torch.equal(pr_e, r_e.cpu()) and torch.equal(ps_e, s_e.cpu())       # exact recall: bit-identical
    if agree == 1.0 and torch.equal(parts[0]["cent"], hf.centroids.cpu()):
        assert torch.equal(pr_c, r_c.cpu()) and torch.equal(ps_c, s_c.cpu())
    else:                                                                       # centroids differ in the last bits:
        same = (pr_c == r_c.cpu()).all(dim=1).float().mean().item()              # a probe may flip for a few queries
        assert same >= 0.97, same
    per = 700 // world
    for r in range(world):
        s_o, r_o = parts[r]["own"]
        assert torch.equal(r_o, pr_c[r * per:(r + 1) * per])


@pytest.mark.parametrize("N", [2_500_000, 4_200_000, 6_000_000])     # 64 / 128 / 256 sample tiles per list
def test_large_bank_samples_grow_with_the_bank(dev, N):
    """A bank well beyond 1 M rows (6 M x 64 here: 1.5 GB): the prefilter's sample grows with the bank --
    128-row groups in the full scan, more sample tiles per inverted list -- so the candidate lists stay
    inside the refine stage's capacity and neither path raises the overflow flag (with the fixed-size
    samples of the 1 M design every call at this size fell back to the all-fp32 kernels).  Results still
    equal the all-fp32 scan / the fp32 lists bit for bit."""
    from aura_snn_rag_amd import ops
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    D, k = 64, 32
    hf = HippocampalFormation(feature_dim=D, max_memories=N, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                              device="cuda", use_centroid_index=True)
    g = torch.Generator(device=dev).manual_seed(99)
    for r0 in range(0, N, 1 << 20):
        n = min(1 << 20, N - r0)
        hf.bulk_write(torch.randn(n, D, generator=g, device=dev), rebuild=False)
    hf.rebuild_centroids()
    now = float(hf.memory_metadata[0, 1].item()) + 5.0
    pick = torch.randint(0, N, (200,), generator=g, device=dev)
    q = torch.cat([hf.memory_features[pick] + 0.05 * torch.randn(200, D, generator=g, device=dev),
                   torch.randn(100, D, generator=g, device=dev)]).contiguous()
    # full scan through the bf16 shadow
    shadow = hf._ensure_shadow()
    assert shadow is not None
    s2, r2, flag = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N,
                                  shadow=shadow, rho=hf._rho, check_overflow=False, return_flag=True)
    assert int(flag) == 0, f"full scan raised flag {int(flag)} at {N} rows"
    s0, r0 = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N, fp32_scan=True)
    assert torch.equal(r0, r2) and torch.equal(s0, s2)
    assert bool((r2[:200, 0] == pick.to(torch.int32)).all())
    # centroid index through the inverted lists on the two-stage scan
    s_c, r_c = hf.recall_batch(q, k=k, now=now, check_overflow=False, fallback_empty=False)
    ivf = hf._ivf
    assert ivf is not None and ivf.valid
    flag_c = int(ops._overflow_flag(q.device).item())
    assert flag_c & ~ops.KNN_FLAG_NO_CANDIDATES == 0, f"inverted lists raised flag {flag_c} at {N} rows"
    s_m, r_m = ops.knn_search(hf.memory_features, hf._inv_norm, hf.memory_metadata, q, k, now, count=N,
                              centroids=hf.centroids, nprobe=8, fp32_scan=True)
    assert torch.equal(r_m, r_c) and torch.equal(s_m, s_c)
    del hf
    torch.cuda.empty_cache()


def test_recall_with_precomputed_probes(bank_1m, dev):
    """`HippocampalFormation.probe` + `recall_batch(probe_ids=...)` (what a sharded bank does: every query is
    probed once, by the rank that brings it) returns exactly what the self-probing recall returns; the ids
    are the 8 nearest centroid rows in distance order."""
    hf, q, pick, now, host = bank_1m
    ids = hf.probe(q)
    assert ids is not None and ids.dtype == torch.int32 and tuple(ids.shape) == (q.shape[0], 8)
    d = torch.cdist(q[:40].double().cpu(), host["cent"].double())
    ref = torch.topk(d, 8, dim=1, largest=False).indices
    agree = (ids[:40].cpu().long() == ref).float().mean().item()
    assert agree > 0.97, agree                                   # fp32 vs fp64 distances: near-ties may swap
    s0, r0 = hf.recall_batch(q, k=32, now=now)
    s1, r1 = hf.recall_batch(q, k=32, now=now, probe_ids=ids)
    assert torch.equal(r0, r1) and torch.equal(s0, s1)
