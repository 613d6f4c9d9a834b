"""Centroid-index (IVF) recall on a 1M x 768 bank, single GPU: retrievals/s vs query-batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0"); D, k = 768, 32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
bank = torch.empty(N, D, device=dev)
for r0 in range(0, N, 1 << 17):
    bank[r0:r0 + (1 << 17)] = torch.randn(min(1 << 17, N - r0), D, device=dev)
inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9
cent = torch.zeros(256, D, device=dev); cent[:] = bank[torch.randperm(N)[:256].to(dev)]
assign = ops.kmeans_assign(bank, cent, N, 256); ops.kmeans_update(bank, assign, cent, 256)
assign = ops.kmeans_assign(bank, cent, N, 256); meta[:, 2] = assign.float()
order = torch.sort(assign, stable=True).indices.to(torch.int32).contiguous()
lens = torch.bincount(assign.long(), minlength=256)[:256].to(torch.int32).contiguous()
off = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(lens, 0).to(torch.int32)]).contiguous()
longest = int(torch.topk(lens, 8).values.sum().item())
cap = ops.ivf_capacity(longest, k)
print(f"N={N} lists: min {int(lens.min())} max {int(lens.max())} sum-of-8-longest {longest} cap {cap}")
for nq in (256, 1024, 2048, 4096):
    q = torch.randn(nq, D, device=dev)
    s, r, ovf = ops.knn_search_ivf(bank, inv, meta, q, k, 1.7e9, N, cent, 8, order, off, lens, cap)
    if nq == 256:
        s2, r2 = ops.knn_search(bank, inv, meta, q, k, 1.7e9, count=N, centroids=cent, nprobe=8)
        print("identical to masked full scan:", bool(torch.equal(r, r2) and torch.equal(s, s2)), "overflow", int(ovf.item()))
    for _ in range(2): ops.knn_search_ivf(bank, inv, meta, q, k, 1.7e9, N, cent, 8, order, off, lens, cap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.knn_search_ivf(bank, inv, meta, q, k, 1.7e9, N, cent, 8, order, off, lens, cap)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"  nq={nq}: {dt*1e3:.3f} ms/batch -> {nq/dt:,.0f} retrievals/s   (bank bytes/batch {N*D*4/1e9:.2f} GB -> {N*D*4/dt/1e12:.2f} TB/s if read once)")
