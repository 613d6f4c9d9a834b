import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda")
for D, nq, kc in ((32, 1, 16), (32, 37, 16), (64, 5, 256)):
    g = torch.Generator().manual_seed(1)
    cent = torch.zeros(256, D); cent[:kc] = torch.randn(kc, D, generator=g)
    q = torch.randn(nq, D, generator=g)
    ids = ops.centroid_probe(q.to(dev), cent.to(dev), 8).cpu()
    d = torch.cdist(q.double(), cent.double())
    ref = torch.topk(d, 8, dim=1, largest=False).indices
    print(D, nq, kc, "ids", ids[0].tolist(), "ref", ref[0].tolist(), "d(ids)", [round(float(d[0, i]), 4) for i in ids[0]], "d(ref)", [round(float(x), 4) for x in d[0, ref[0]]])
