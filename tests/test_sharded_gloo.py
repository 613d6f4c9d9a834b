"""world_size-2 gloo test of the row-sharded recall plumbing: per-shard
top-k -> all_gather -> merge must equal the single-process oracle on the whole bank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import aura_oracle as O
from tests import cpu_stub_ops as stub

NOW = 1.7e9


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make(N=3000, D=24, nq=10):
    g = torch.Generator().manual_seed(7)
    bank = torch.randn(N, D, generator=g)
    meta = torch.zeros(N, 4); meta[:, 0] = 0.5 + 0.5 * torch.rand(N, generator=g); meta[:, 1] = NOW
    q = bank[torch.randint(0, N, (nq,), generator=g)] + 0.05 * torch.randn(nq, D, generator=g)
    return bank, meta, q


def _worker(rank, world, port, out):
    from aura_snn_rag_amd.sharded import ShardedRecall, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bank, meta, q = _make()
    r0, r1 = shard_rows(bank.shape[0], world, rank)
    sb, sm = bank[r0:r1].contiguous(), meta[r0:r1].contiguous()
    inv = torch.empty(r1 - r0)

    def local(qq, k):
        return stub.knn_search(sb, inv, sm, qq, k, NOW, idx_base=r0)
    rec = ShardedRecall(local, stub.topk_merge)
    s1, i1 = rec.recall(q, 8)                                   # replicated queries
    myq = q[rank * 5:(rank + 1) * 5].contiguous()               # each rank owns 5 queries
    s2, i2 = rec.recall(myq, 8, all_gather_queries=True)
    if rank == 0:
        torch.save((s1, i1), out + ".rep")
    torch.save((s2, i2), out + f".own{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_cover_exactly():
    from aura_snn_rag_amd.sharded import shard_rows
    for total, world in ((100_000, 8), (10, 3), (7, 8), (1_000_000, 8)):
        spans = [shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_two_rank_sharded_recall_matches_single_process(tmp_path):
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    bank, meta, q = _make()
    ri, rs = O.knn_exact_batch(bank, meta[:, 0], meta[:, 1], q, 8, NOW)
    s1, i1 = torch.load(out + ".rep")
    assert torch.equal(i1.long(), ri) and torch.allclose(s1, rs, atol=1e-6)
    for rank in range(2):
        s2, i2 = torch.load(out + f".own{rank}")
        assert torch.equal(i2.long(), ri[rank * 5:(rank + 1) * 5])


# ----------------------------------------------------------------------------------------------
# Sharded writes (owner by slot), the centroid rebuild as partial sums + all_reduce, and recall in
# candidate mode over the shards -- against ONE bank holding everything.
# ----------------------------------------------------------------------------------------------
def _clustered_rows(n, D, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(12, D, generator=g) * 4
    return c[torch.randint(0, 12, (n,), generator=g)] + 0.4 * torch.randn(n, D, generator=g)


def _bank_kw(D, M):
    return dict(n_place_cells=4, n_time_cells=3, n_grid_cells=3, max_memories=M, feature_dim=D, device="cpu",
                use_centroid_index=True, overflow="fifo")


def _run_single_bank(feats, q, perm, D, M, interval, ck, extra):
    """The same sequence of operations on one HippocampalFormation holding all rows."""
    from aura_snn_rag_amd.core import hippocampal as H
    H.ops = stub
    H.time.time = lambda: NOW
    hf = H.HippocampalFormation(**_bank_kw(D, M))
    hf.centroids_k, hf.centroids_update_interval = ck, 10 ** 9     # rebuilds are explicit here
    n0 = feats.shape[0]
    hf.create_episodic_memories([f"m{i}" for i in range(n0)], feats)
    hf.rebuild_centroids(perm=perm)
    hf.create_episodic_memories([f"x{i}" for i in range(extra.shape[0])], extra)     # online centroid updates
    s_c, r_c = hf.recall_batch(q, k=6, now=NOW)
    s_e, r_e = hf.recall_batch(q, k=6, now=NOW, use_candidates=False)
    return hf, (s_c, r_c, s_e, r_e)


def _sharded_worker(rank, world, port, out):
    from aura_snn_rag_amd.core import hippocampal as H
    from aura_snn_rag_amd.sharded import ShardedHippocampus
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H.ops = stub
    H.time.time = lambda: NOW
    D, M, ck = 16, 1200, 16
    feats, extra = _clustered_rows(900, D, 1), _clustered_rows(350, D, 2)      # 900 + 350 > M: ring overwrites
    q = _clustered_rows(14, D, 3)
    perm = torch.randperm(900, generator=torch.Generator().manual_seed(4))
    local = H.HippocampalFormation(**_bank_kw(D, M // world))
    local.centroids_k, local.centroids_update_interval = ck, 10 ** 9
    sh = ShardedHippocampus(local, M, ops_module=stub, now_fn=lambda: NOW)
    for i in range(0, 900, 250):                                # batches that straddle the shard boundary
        sh.write([f"m{j}" for j in range(i, min(i + 250, 900))], feats[i:i + 250])
    assert sh.memory_count == 900 and local.memory_count == (600 if rank == 0 else 300)
    sh.rebuild_centroids(perm=perm)
    sh.write([f"x{i}" for i in range(350)], extra)
    assert sh.memory_count == M and local.memory_count == 600 and sh._write_cursor == 50
    s_c, r_c = sh.recall_batch(q, k=6, now=NOW)
    s_e, r_e = sh.recall_batch(q, k=6, now=NOW, use_candidates=False)
    myq = q[rank * 7:(rank + 1) * 7].contiguous()
    s_o, r_o = sh.recall_batch(myq, k=6, now=NOW, all_gather_queries=True)
    torch.save(dict(feats=local.memory_features.clone(), meta=local.memory_metadata.clone(), cent=local.centroids.clone(),
                    counts=local.centroid_counts.clone(), res=(s_c, r_c, s_e, r_e), own=(s_o, r_o),
                    ids=dict(local.id_to_idx)), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_writes_rebuild_and_candidate_recall(tmp_path, monkeypatch):
    out = str(tmp_path / "sh")
    mp.spawn(_sharded_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from aura_snn_rag_amd.core import hippocampal as H
    monkeypatch.setattr(H, "ops", stub)
    monkeypatch.setattr(H.time, "time", lambda: NOW)
    D, M, ck = 16, 1200, 16
    feats, extra = _clustered_rows(900, D, 1), _clustered_rows(350, D, 2)
    q = _clustered_rows(14, D, 3)
    perm = torch.randperm(900, generator=torch.Generator().manual_seed(4))
    hf, (s_c, r_c, s_e, r_e) = _run_single_bank(feats, q, perm, D, M, 10 ** 9, ck, extra)
    parts = [torch.load(out + f".{r}") for r in range(2)]
    # owner-by-slot routing: the two shards side by side ARE the single bank
    assert torch.equal(torch.cat([p["feats"] for p in parts]), hf.memory_features)
    meta = torch.cat([p["meta"] for p in parts])
    assert torch.equal(meta[:, :2], hf.memory_metadata[:, :2])
    # centroid table: replicated, equal on both ranks, and equal to the single bank's up to the
    # summation order of the all-reduced partial sums
    assert torch.equal(parts[0]["cent"], parts[1]["cent"]) and torch.equal(parts[0]["counts"], parts[1]["counts"])
    assert torch.allclose(parts[0]["cent"], hf.centroids, rtol=1e-5, atol=1e-5)
    assert torch.equal(parts[0]["counts"], hf.centroid_counts)
    assert torch.equal(meta[:, 2], hf.memory_metadata[:, 2]), "centroid ids of the sharded and the single bank differ"
    ids = {}
    for r, p in enumerate(parts):
        ids.update({k_: v + r * (M // 2) for k_, v in p["ids"].items()})
    assert ids == hf.id_to_idx
    # recall: candidate mode and exact, replicated queries and per-rank query blocks
    ps_c, pr_c, ps_e, pr_e = parts[0]["res"]
    assert torch.equal(pr_e.long(), r_e.long()) and torch.allclose(ps_e, s_e, atol=1e-6)
    assert torch.equal(pr_c.long(), r_c.long()) and torch.allclose(ps_c, s_c, atol=1e-6)
    for r in range(2):
        s_o, r_o = parts[r]["own"]
        assert torch.equal(r_o.long(), r_c[r * 7:(r + 1) * 7].long())


# ----------------------------------------------------------------------------------------------
# Seeding path of the sharded bank (BASELINE config 5): every rank bulk-writes ITS OWN rows, one
# collective rebuild, recall -- against ONE bank holding the shards side by side; the shard-fill
# check of write() after bulk_write; persistence of a sharded bank.
# ----------------------------------------------------------------------------------------------
def _bulk_worker(rank, world, port, out):
    from aura_snn_rag_amd.core import hippocampal as H
    from aura_snn_rag_amd.sharded import ShardedHippocampus
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H.ops = stub
    H.time.time = lambda: NOW
    D, R, ck = 16, 500, 16
    rows = _clustered_rows(2 * R, D, 11)
    q = _clustered_rows(12, D, 12)
    perm = torch.randperm(2 * R, generator=torch.Generator().manual_seed(13))
    local = H.HippocampalFormation(**_bank_kw(D, R))
    local.centroids_k, local.centroids_update_interval = ck, 10 ** 9
    sh = ShardedHippocampus(local, 2 * R, ops_module=stub, now_fn=lambda: NOW)
    # a partly filled pair of shards first: write() must refuse it ON EVERY RANK (300 + 300 rows: rank 1 should
    # hold none of the first 600 slots)
    part = ShardedHippocampus(H.HippocampalFormation(**_bank_kw(D, R)), 2 * R, ops_module=stub, now_fn=lambda: NOW)
    part.local.centroids_k = ck
    assert part.bulk_write(rows[rank * 300:(rank + 1) * 300]) == 600
    raised = False
    try:
        part.write(["a", "b"], rows[:2])
    except RuntimeError:
        raised = True
    # the seeding path proper: full shards
    assert sh.bulk_write(rows[rank * R:(rank + 1) * R]) == 2 * R
    sh.rebuild_centroids(perm=perm)
    s_c, r_c = sh.recall_batch(q, k=6, now=NOW)
    s_e, r_e = sh.recall_batch(q, k=6, now=NOW, use_candidates=False)
    cent0, counts0 = local.centroids.clone(), local.centroid_counts.clone()
    # a write after the full seeding: the ring overwrites global slots 0.. (consistent layout: no error)
    sh.write([f"w{i}" for i in range(5)], rows[:5] + 0.01)
    # persistence: state dict + bank_state into a fresh pair of objects, same recall
    sd, bs = {k_: v.clone() for k_, v in local.state_dict().items()}, sh.bank_state()
    local2 = H.HippocampalFormation(**_bank_kw(D, R))
    local2.centroid_counts = torch.zeros(ck)                    # (this test shrinks centroids_k: the counts buffer follows)
    local2.load_state_dict(sd)
    sh2 = ShardedHippocampus(local2, 2 * R, ops_module=stub, now_fn=lambda: NOW)
    sh2.load_bank_state(bs)
    s_a, r_a = sh.recall_batch(q, k=6, now=NOW)
    s_b, r_b = sh2.recall_batch(q, k=6, now=NOW)
    torch.save(dict(raised=raised, res=(s_c, r_c, s_e, r_e), cent=cent0, counts=counts0,
                    meta=local.memory_metadata.clone(), reload_equal=bool(torch.equal(r_a, r_b) and torch.equal(s_a, s_b)),
                    count2=sh2.memory_count, cursor2=sh2._write_cursor, cursor=sh._write_cursor,
                    id5=local2.id_of_row(4) if rank == 0 else None), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bulk_write_rebuild_recall_and_persistence(tmp_path, monkeypatch):
    out = str(tmp_path / "bulk")
    mp.spawn(_bulk_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from aura_snn_rag_amd.core import hippocampal as H
    monkeypatch.setattr(H, "ops", stub)
    monkeypatch.setattr(H.time, "time", lambda: NOW)
    D, R, ck = 16, 500, 16
    rows = _clustered_rows(2 * R, D, 11)
    q = _clustered_rows(12, D, 12)
    perm = torch.randperm(2 * R, generator=torch.Generator().manual_seed(13))
    hf = H.HippocampalFormation(**_bank_kw(D, 2 * R))
    hf.centroids_k, hf.centroids_update_interval = ck, 10 ** 9
    hf.bulk_write(rows, rebuild=False)
    hf.rebuild_centroids(perm=perm)
    s_c, r_c = hf.recall_batch(q, k=6, now=NOW)
    s_e, r_e = hf.recall_batch(q, k=6, now=NOW, use_candidates=False)
    parts = [torch.load(out + f".{r}") for r in range(2)]
    assert all(p["raised"] for p in parts), "write() after an uneven bulk_write must raise on every rank"
    assert torch.equal(parts[0]["cent"], parts[1]["cent"]) and torch.equal(parts[0]["counts"], parts[1]["counts"])
    assert torch.allclose(parts[0]["cent"], hf.centroids, rtol=1e-5, atol=1e-5)
    ps_c, pr_c, ps_e, pr_e = parts[0]["res"]
    assert torch.equal(pr_e.long(), r_e.long()) and torch.allclose(ps_e, s_e, atol=1e-6)
    assert torch.equal(pr_c.long(), r_c.long()) and torch.allclose(ps_c, s_c, atol=1e-6)
    for p in parts:
        assert p["reload_equal"] and p["count2"] == 2 * R and p["cursor2"] == p["cursor"] == 5
    assert parts[0]["id5"] == "w4"                               # the ring overwrote global slots 0..4 (rank 0's)
