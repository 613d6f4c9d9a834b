"""Timing probe: ops.centroid_probe (no list fill) and the ivf2 recall's probe launch at 2048 x 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
from bench import timed_events
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for nq in (256, 2048, 16384):
    q = torch.randn(nq, 768, generator=g, device=dev)
    c = torch.randn(256, 768, generator=g, device=dev) * 0.3
    ms = timed_events(lambda: ops.centroid_probe(q, c, 8), iters=20, warm=3)
    print(f"{os.environ.get('AURA_HIP_LIB', 'base')}: centroid_probe nq={nq}: {ms * 1e3:.1f} us", flush=True)
