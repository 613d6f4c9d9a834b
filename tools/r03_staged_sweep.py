"""Random sweep: the staged recall (both exchanges, each bank's own bounds) against the unstaged one -- rows and
score bits must be equal for every shape.  python tools/r03_staged_sweep.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
dev = torch.device("cuda")
g = torch.Generator().manual_seed(5)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 14
bad = 0
for c in range(cases):
    D = int(torch.randint(1, 97, (1,), generator=g)) * 8
    N = int(torch.randint(9000, 90000, (1,), generator=g))
    nq = [1, 7, 300, 2500, 9000][int(torch.randint(0, 5, (1,), generator=g))]
    k = [1, 5, 20, 64, 150][int(torch.randint(0, 5, (1,), generator=g))]
    parts = [1, 2, 4, 8][int(torch.randint(0, 4, (1,), generator=g))]
    clustered = bool(torch.randint(0, 2, (1,), generator=g))
    feats = torch.randn(N, D, generator=g)
    if clustered:
        cen = torch.randn(40, D, generator=g) * 3
        feats = feats * 0.3 + cen[torch.randint(0, 40, (N,), generator=g)]
    hf = HippocampalFormation(feature_dim=D, max_memories=N, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                              device="cuda", use_centroid_index=True)
    hf.bulk_write(feats.to(dev), rebuild=True)
    now = float(hf.memory_metadata[0, 1].item()) + 3.0
    q = (feats[torch.randint(0, N, (nq,), generator=g)] + 0.1 * torch.randn(nq, D, generator=g)).to(dev).contiguous()
    s0, r0 = hf.recall_batch(q, k=k, now=now)
    calls = []

    def fn(b):
        calls.append(tuple(b.shape))
        return b[:, 0].contiguous()
    s1, r1 = hf.recall_batch(q, k=k, now=now, bound_exchange=(fn, parts))
    ok = torch.equal(r0, r1) and torch.equal(s0, s1)
    bad += 0 if ok else 1
    print(f"case {c}: N {N} D {D} nq {nq} k {k} parts {parts} clustered {clustered}: exchanges {len(calls)} "
          f"{'ok' if ok else 'MISMATCH: ' + str(int((r0 != r1).sum())) + ' entries'}", flush=True)
    del hf
    torch.cuda.empty_cache()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
