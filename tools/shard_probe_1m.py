"""Per-rank local-scan time of BASELINE config 4 emulated on one GPU: a 1M x 768 bank row-sharded over
8 ranks (125k rows per rank), every rank scanning its shard for all 8 x 256 all-gathered queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0"); D, k = 768, 32
for world, total in ((8, 1_000_000), (1, 1_000_000)):
    N, nq = total // world, 256 * world
    bank = torch.randn(N, D, device=dev); inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
    meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9; meta[:, 2] = -1
    shadow = torch.empty(N, D, dtype=torch.bfloat16, device=dev); ops.bank_shadow_update(bank, shadow)
    q = torch.randn(nq, D, device=dev)
    for use_sh in (True, False):
        sh = shadow if use_sh else None
        for _ in range(3): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False, shadow=sh)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False, shadow=sh)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"total {total} rows, world={world}, shadow={use_sh}: shard {N} rows x {nq} queries: {dt*1e3:.3f} ms/step -> "
              f"{nq/dt:,.0f} query-shard results/s per rank; x{world} ranks at 256 own queries each = {256*world/dt:,.0f} retrievals/s aggregate (compute only)")
    del bank, inv, meta, shadow, q
