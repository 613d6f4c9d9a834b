"""Neuron-loop throughput of the loaded library (AURA_HIP_LIB selects a variant build)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
r = bench.secondary_neurons(dev)
print(os.environ.get("AURA_HIP_LIB", "default"),
      {k: (round(v["ms"], 4), round(v.get("hbm_frac_of_8TBs", 0), 3)) for k, v in r.items()}, flush=True)
