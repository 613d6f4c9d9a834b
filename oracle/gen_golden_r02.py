#!/usr/bin/env python3
"""Round-2 additions to the golden vectors, generated FROM THE REFERENCE ITSELF (see gen_golden.py):
    python oracle/gen_golden_r02.py
rate_codes.pt -- place / grid / time-cell rate codes of HippocampalFormation (hippocampal.py:120-193)
for seeded cell parameters, three locations and three elapsed times."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from oracle import _ref_loader as L

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    H = L.load("core.hippocampal")
    clock = [1.7e9]
    H.time.time = lambda: clock[0]
    torch.manual_seed(123)
    hf = H.HippocampalFormation(spatial_dimensions=2, n_place_cells=200, n_time_cells=40, n_grid_cells=60,
                                max_memories=8, feature_dim=8, device="cpu")
    cases = []
    for loc, dt in (([0.0, 0.0], 0.0), ([3.25, -1.5], 2.5), ([-7.75, 9.0], 180.0)):
        hf.update_spatial_state(torch.tensor(loc))
        sp = hf.get_spatial_context()
        clock[0] = 1.7e9 + dt
        tc = hf.get_temporal_context()
        cases.append(dict(location=torch.tensor(loc), elapsed=dt, place=sp["place_cells"].clone(),
                          grid=sp["grid_cells"].clone(), time=tc["time_cells"].clone()))
    torch.save(dict(seed=123, ctor=dict(spatial_dimensions=2, n_place_cells=200, n_time_cells=40, n_grid_cells=60,
                                        max_memories=8, feature_dim=8),
                    buffers={k: v.clone() for k, v in hf.state_dict().items() if k.startswith(("place_", "grid_", "time_", "k_const"))},
                    cases=cases), os.path.join(OUT, "rate_codes.pt"))
    print("rate_codes.pt", os.path.getsize(os.path.join(OUT, "rate_codes.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
