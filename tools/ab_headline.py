"""In-process A/B of AURA_CS_DBG flag sets on the headline workload (1 M x 768 rows, 2048 queries, k = 32).

    python tools/ab_headline.py 0 4096 16384 [--reps 7] [--steps 40] [--exact]

One bank, one plan, one process: the flag sets are alternated `reps` times and the medians reported (step time by
HIP events around `steps` calls; dominant-kernel time from the library's own event pairs).  Successive processes on
one box differ by +-5 %, which hides most of the effects the flags are for."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("flags", nargs="+", type=int)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=2048)
    ap.add_argument("--exact", action="store_true")
    a = ap.parse_args()
    from aura_snn_rag_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    D, k = 768, 32
    hf = bench.new_bank(a.rows, D, dev)
    bench.fill_bank(hf, a.rows, D, 1234, dev)
    hf.rebuild_centroids(perm=torch.randperm(a.rows, generator=torch.Generator().manual_seed(7)))
    now = float(hf.memory_metadata[0, 1].item())
    g = torch.Generator(device=dev).manual_seed(99)
    pick = torch.randint(0, a.rows, (a.nq // 2,), generator=g, device=dev)
    q = torch.cat([hf.memory_features[pick] + 0.05 * torch.randn(a.nq // 2, D, generator=g, device=dev),
                   torch.randn(a.nq - a.nq // 2, D, generator=g, device=dev)]).contiguous()

    def step():
        return hf.recall_batch(q, k=k, now=now, use_candidates=not a.exact)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    res = {f: ([], []) for f in a.flags}
    for rep in range(a.reps):
        for f in (a.flags if rep % 2 == 0 else a.flags[::-1]):
            lib.aura_debug_cs_flags(f)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            lib.aura_profile_begin(a.steps * 16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            buf = (ctypes.c_float * (a.steps * 16))()
            n = lib.aura_profile_end(buf, a.steps * 16)
            res[f][0].append(e0.elapsed_time(e1) / a.steps)
            res[f][1].append(sum(buf[j] for j in range(n)) / max(n, 1))
    lib.aura_debug_cs_flags(0)
    clk = torch.zeros(1, device=dev)
    lib.aura_debug_clock_mhz(ctypes.c_void_p(clk.data_ptr()), 2000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print(f"shader clock (s_memtime / s_memrealtime over 2 ms, after the runs): {float(clk.item()):.0f} MHz")
    for f in a.flags:
        st, km = res[f]
        print(f"flags {f:6d}: step median {statistics.median(st):.4f} ms (min {min(st):.4f} max {max(st):.4f}); "
              f"dominant kernel median {statistics.median(km):.4f} ms (min {min(km):.4f} max {max(km):.4f})")


if __name__ == "__main__":
    main()
