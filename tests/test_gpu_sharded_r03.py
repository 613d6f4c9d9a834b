"""the sharded bank's collectives over RCCL ("nccl") on hardware -- a world of ONE rank
with `force_collectives`, so that the int32 bit-cast packing, aura_topk_merge, the centroid all-reduce, the
permutation broadcast and the bound exchange all run through RCCL on the one-GPU test box -- and the bound
exchange itself (staged inverted-list recall == the unstaged one, bit for bit)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(D=64, n=30000, nq=900, seed=5):
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(200, D, generator=g) * 3
    feats = centres[torch.randint(0, 200, (n,), generator=g)] + torch.randn(n, D, generator=g)
    extra = centres[torch.randint(0, 200, (400,), generator=g)] + torch.randn(400, D, generator=g)
    q = centres[torch.randint(0, 200, (nq,), generator=g)] + torch.randn(nq, D, generator=g)
    return feats, extra, q, torch.randperm(n, generator=g)


def _rccl_worker(rank, port, out):
    import torch.distributed as dist
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    from aura_snn_rag_amd.sharded import ShardedHippocampus
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    D, M = 64, 40000
    feats, extra, q, perm = _data(D)
    kw = dict(feature_dim=D, max_memories=M, n_place_cells=8, n_time_cells=4, n_grid_cells=4, device="cuda",
              use_centroid_index=True, overflow="fifo")
    now = 1.7e9
    res = {}
    trace = os.environ.get("AURA_TEST_TRACE") is not None

    def mark(what):                                                 # AURA_TEST_TRACE: localise a device fault
        if trace:
            torch.cuda.synchronize()
            print(f"[rccl-test] ok: {what}", file=sys.stderr, flush=True)
    for name, force in (("plain", False), ("rccl", True)):
        local = HippocampalFormation(**kw)
        local.centroids_update_interval = 10 ** 9
        sh = ShardedHippocampus(local, M, now_fn=lambda: now, force_collectives=force)
        sh.bulk_write(feats[:20000].to(dev)); mark(f"{name} bulk_write")
        for i in range(20000, 30000, 4000):
            sh.write([f"m{j}" for j in range(i, min(i + 4000, 30000))], feats[i:i + 4000])
        mark(f"{name} writes")
        sh.rebuild_centroids(perm=None if force else perm)          # forced: rank 0's draw is BROADCAST ...
        mark(f"{name} rebuild")
        if force:
            sh.rebuild_centroids(perm=perm)                          # ... then the same perm as the plain run
        sh.write([f"x{j}" for j in range(400)], extra)               # replicated online centroid update
        mark(f"{name} online write")
        local.memory_metadata[:, 1] = now                            # bulk_write stamps the wall clock: one clock for both runs
        s_c, r_c = sh.recall_batch(q.to(dev), k=9, now=now, all_gather_queries=True)
        mark(f"{name} candidate recall")
        s_e, r_e = sh.recall_batch(q.to(dev), k=9, now=now, use_candidates=False)
        mark(f"{name} exact recall")
        res[name] = dict(s_c=s_c.cpu(), r_c=r_c.cpu(), s_e=s_e.cpu(), r_e=r_e.cpu(), cent=local.centroids.cpu(),
                         counts=local.centroid_counts.cpu(), meta=local.memory_metadata[:, 2].cpu(),
                         exchanges=sh.exchanges, agreed=sh._exch_ok)
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_over_rccl_world_of_one(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    a, b = res["plain"], res["rccl"]
    assert b["exchanges"] >= 1 and b["agreed"], "the bound exchange did not run over RCCL"
    assert a["exchanges"] == 0
    # all-reduce(SUM) over one rank is the identity: everything is bit-identical to the run without collectives
    assert torch.equal(a["cent"], b["cent"]) and torch.equal(a["counts"], b["counts"]) and torch.equal(a["meta"], b["meta"])
    assert torch.equal(a["r_e"], b["r_e"]) and torch.equal(a["s_e"], b["s_e"])
    assert torch.equal(a["r_c"], b["r_c"]) and torch.equal(a["s_c"], b["s_c"])


def test_staged_inverted_list_recall_equals_unstaged(dev):
    """aura_knn_search_ivf2_staged: stage 1 -> bounds, stage 2 with (a) the bank's own bound, (b) a TIGHTER valid
    bound (the true k-th best score minus a hair: what other shards may contribute), (c) a useless bound: rows and
    score bits always equal the unstaged recall; the tighter bound shrinks the candidate lists."""
    from aura_snn_rag_amd import ops
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    D = 96
    feats, _, q, perm = _data(D, n=60000, nq=2500, seed=9)
    hf = HippocampalFormation(feature_dim=D, max_memories=60000, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                              device="cuda", use_centroid_index=True)
    hf.bulk_write(feats.to(dev), rebuild=False)
    hf.rebuild_centroids(perm=perm)
    now = float(hf.memory_metadata[0, 1].item())
    qd = q.to(dev).contiguous()
    s0, r0 = hf.recall_batch(qd, k=20, now=now)
    ivf = hf._ivf
    assert ivf is not None and ivf.valid
    kth = s0[:, -1].clone()
    kth[r0[:, -1] < 0] = -3.0e38                                  # fewer than k candidates: no bound from outside
    counts = {}
    # (column 1, the k2-th largest bound, is only valid as the MINIMUM over `parts` disjoint shards: one bank alone
    #  may use column 0 only)
    for name, make in (("own", lambda b: b[:, 0]),
                       ("tight", lambda b: torch.maximum(b[:, 0], kth - 1e-4)),
                       ("none", lambda b: torch.full_like(b[:, 0], -3.0e38))):
        calls = []

        def fn(b, make=make):
            calls.append(b.shape[0])
            return make(b).contiguous()
        s1, r1 = hf.recall_batch(qd, k=20, now=now, bound_exchange=(fn, 4))
        assert calls == [2500, 2500], calls                        # sampled bounds, then the candidates' bounds
        assert torch.equal(r1, r0) and torch.equal(s1, s0), name
    # a bound from outside ABOVE the bank's own k-th best score (other shards hold better rows): the second exchange
    # hands it to the refine, which returns the rows that can still reach it -- a prefix of the unstaged result, at
    # least the ten rows that score at least the bound, -1 behind it -- and re-scores far fewer candidates
    b10 = s0[:, 9].clone()
    state = {"n": 0}

    def fn10(b):
        state["n"] += 1
        return (b[:, 0] if state["n"] == 1 else torch.maximum(b[:, 0], b10)).contiguous()
    s2, r2 = hf.recall_batch(qd, k=20, now=now, bound_exchange=(fn10, 4))
    assert state["n"] == 2
    valid = r2 >= 0
    # the ten rows that score at least the bound survive, in place; behind them come the rows whose UPPER bound
    # still reaches it (a subset of the unstaged result, same score bits, ranked among themselves), then -1
    assert torch.equal(r2[:, :10], r0[:, :10]) and torch.equal(s2[:, :10], s0[:, :10])
    # (a survivor below the bound need not be one of the bank's own top 20 -- 20 better rows exist -- but it scores
    #  below the bound, i.e. below the global k-th best, and never reaches the merged result)
    below = s2[:, 10:][valid[:, 10:]]
    assert bool((below <= b10.unsqueeze(1).expand(-1, 10)[valid[:, 10:]]).all())
    sv = torch.where(valid, s2, torch.full_like(s2, -3.0e38))
    assert bool((sv[:, :-1] >= sv[:, 1:]).all()), "sorted, valid entries first"
    assert float(valid.float().sum(dim=1).mean()) < 19.5, "the external bound did not prune"
    # staged passes of 8192: 9000 queries -> two exchanges
    q9 = torch.cat([qd, qd, qd, qd[:1500]]).contiguous()
    calls = []

    def fn9(b):
        calls.append(b.shape[0])
        return b[:, 0].contiguous()
    s9, r9 = hf.recall_batch(q9, k=20, now=now, bound_exchange=(fn9, 1))
    assert calls == [8192, 8192, 808, 808]
    assert torch.equal(r9[:2500], r0) and torch.equal(s9[7500:], s0[:1500])
    # another path (exact recall) still performs the exchanges, with neutral bounds
    calls = []
    hf.recall_batch(qd, k=20, now=now, use_candidates=False, bound_exchange=(fn9, 1))
    assert calls == [2500, 2500]
