import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
from tools.coarse_probe import bank_of
dev = torch.device("cuda:0")
N, D, nq, k = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 768, 256, 32
bank, inv, meta = bank_of(N, D, dev)
g = torch.Generator().manual_seed(1)
q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.5 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
now = 1.7e9 + 100.0
s0, i0 = ops.knn_search(bank, inv, meta, q, k, now, fp32_scan=True)
s1, i1 = ops.knn_search(bank, inv, meta, q, k, now, check_overflow=False)
bad = (i0 != i1).any(1).nonzero().flatten().tolist()
print("D", D, "bad queries:", bad)
for b in bad[:6]:
    miss = [r for r in i0[b].tolist() if r not in i1[b].tolist()]
    pos = [(i0[b] == r).nonzero().item() for r in miss]
    print(b, "missing rows", miss, "ranks", pos, "row%16", [r % 16 for r in miss], "tile", [r // 16 for r in miss])
