"""Dev probe: centroid-candidate recall at config 2 through the two-stage scan (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0"); N, D, nq, k = 100000, 768, 256, 32
g = torch.Generator().manual_seed(0)
bank = torch.randn(N, D, generator=g).to(dev); inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9
cent = bank[torch.randperm(N, generator=g)[:256].to(dev)].clone()
meta[:, 2] = ops.kmeans_assign(bank, cent, N, 256).float()
sh = torch.empty(N, D, dtype=torch.bfloat16, device=dev); ops.bank_shadow_update(bank, sh)
q = (bank[:nq] + 0.05 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
for _ in range(20):
    ops.knn_search(bank, inv, meta, q, k, 1.7e9, centroids=cent, nprobe=8, shadow=sh, check_overflow=False)
torch.cuda.synchronize()
