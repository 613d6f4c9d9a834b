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
