"""Timing probe (GPU box): the online centroid update at 1M x 768 -- serial vs parallel kernel, and the
MemoryAugmentedLayer store+retrieve pattern.  python tools/r03_write_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
from bench import new_bank, fill_bank, timed_wall

dev = torch.device("cuda")
N, D = 1_000_000, 768
hf = new_bank(N + 8192, D, dev)
fill_bank(hf, N, D, 1234, dev)
torch.manual_seed(7)
hf.rebuild_centroids()
hf.centroids_update_interval = 10 ** 9          # no rebuild inside the timed writes
hf._overflow = 'fifo'
now = float(hf.memory_metadata[0, 1].item())
for B in (8, 256, 512):
    x = torch.randn(B, D, device=dev)
    slots = torch.arange(N, N + B, device=dev)
    cur = hf.current_location.to(dev).float().contiguous()
    for serial in (True, False):
        def w():
            ops.bank_write(hf.memory_features, hf.memory_locations, hf.memory_metadata, hf._inv_norm, x, slots, cur, now,
                           centroids=hf.centroids, centroid_counts=hf.centroid_counts, eff_k=256, distinct_slots=True,
                           serial=serial)
        dt = timed_wall(w, 10, warm=2)
        print(f"B={B} {'serial' if serial else 'online'}: {dt * 1e3:.3f} ms per batch, {dt / B * 1e6:.2f} us per row", flush=True)
q = torch.randn(2048, D, device=dev)
hf.recall_batch(q, k=32, now=now)
for B in (8, 256):
    qb = q[:B].contiguous(); rows = torch.randn(B, D, device=dev); ids = [f"x{j}" for j in range(B)]
    def fwd():
        hf.create_episodic_memories(ids, rows)
        return hf.recall_batch(qb, k=5, now=now)
    dt = timed_wall(fwd, 20)
    print(f"interleaved store+retrieve B={B}: {dt * 1e3:.3f} ms per forward", flush=True)
