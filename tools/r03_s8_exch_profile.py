"""One rank's share of the 8-rank layout with the EMULATED two-step bound exchange (see r03_shard_share.py): run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import new_bank, fill_bank
dev = torch.device("cuda")
D, k, S = 768, 32, 8
rows = 1_000_000 // S
hf = new_bank(rows, D, dev)
fill_bank(hf, rows, D, 1234, dev)
torch.manual_seed(7)
hf.rebuild_centroids()
now = float(hf.memory_metadata[0, 1].item())
q = torch.randn(2048 * S, D, generator=torch.Generator(device=dev).manual_seed(99), device=dev)
ids = hf.probe(q)
fn = lambda b: torch.maximum(b[:, 0], b[:, 1]).contiguous()
for _ in range(12):
    hf.recall_batch(q, k=k, now=now, probe_ids=ids, fallback_empty=False, bound_exchange=(fn, S))
torch.cuda.synchronize()
