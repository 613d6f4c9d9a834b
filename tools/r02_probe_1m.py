"""Dev probe (round 2): the product API at the metric's configuration -- a 1M x 768 bank on one GPU.
bulk_write -> rebuild_centroids -> recall_batch (centroid-index and exact), timings per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd.core.hippocampal import HippocampalFormation

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 1_000_000)); D = 768; k = 32
nqs = [int(x) for x in os.environ.get("NQ", "256,2048").split(",")]
torch.manual_seed(0)
hf = HippocampalFormation(feature_dim=D, max_memories=N, n_place_cells=8, n_time_cells=4, n_grid_cells=4,
                          device="cuda", use_centroid_index=True)
g = torch.Generator().manual_seed(1234)
t0 = time.perf_counter()
for r0 in range(0, N, 1 << 17):
    n = min(1 << 17, N - r0)
    hf.bulk_write(torch.randn(n, D, generator=g).to(dev), rebuild=False)
torch.cuda.synchronize()
print(f"bulk_write {N} rows: {time.perf_counter() - t0:.2f} s (incl. host randn)", flush=True)
for i in range(2):
    t0 = time.perf_counter(); hf.rebuild_centroids(); torch.cuda.synchronize()
    print(f"rebuild_centroids: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
now = float(hf.memory_metadata[0, 1].item())


def timed(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


for nq in nqs:
    q = (hf.memory_features[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.5 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
    for cand in (True, False):
        for chk in (True, False):
            dt = timed(lambda: hf.recall_batch(q, k=k, now=now, use_candidates=cand, check_overflow=chk))
            print(f"nq={nq} candidates={cand} check_overflow={chk}: {dt * 1e3:.3f} ms -> {nq / dt:,.0f} retrievals/s", flush=True)
    sc, rc = hf.recall_batch(q, k=k, now=now, use_candidates=True)
    se, re_ = hf.recall_batch(q, k=k, now=now, use_candidates=False)
    hit = (rc.unsqueeze(2) == re_.unsqueeze(1)).any(2).float().mean().item()
    print(f"nq={nq}: recall@{k} of the centroid-index path vs exact = {hit:.3f}; top-1 agreement {(rc[:, 0] == re_[:, 0]).float().mean().item():.3f}", flush=True)
# write -> recall interleave (MemoryAugmentedLayer pattern: store B rows, retrieve B queries per forward)
B = 8
q = torch.randn(B, D, device=dev)
def fwd():
    hf._overflow = 'fifo'
    hf.create_episodic_memories([f"x{i}" for i in range(B)], torch.randn(B, D, device=dev))
    return hf.recall_batch(q, k=5, now=now)
dt = timed(fwd, it=5)
print(f"interleaved store({B}) + retrieve({B}) at {N} rows: {dt * 1e3:.2f} ms per forward", flush=True)
