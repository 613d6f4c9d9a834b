"""Per-rank local-scan time of `bench.py --gpus N` emulated on one GPU: N ranks share a
100k-row bank, so a rank scans 100k/N rows for N*256 all-gathered queries (constant FLOP)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0"); D, k = 768, 32
for world in (1, 2, 4, 8):
    N, nq = 100_000 // world, 256 * world
    bank = torch.randn(N, D, device=dev); inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
    meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9; meta[:, 2] = -1
    q = torch.randn(nq, D, device=dev)
    for _ in range(3): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"world={world}: shard {N} rows x {nq} queries: {dt*1e3:.3f} ms/step  -> {nq/dt:,.0f} retrievals/s per rank-step")
