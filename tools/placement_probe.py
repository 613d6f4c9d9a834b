"""Does the filter launch's time depend on where the bank's tensors land in memory?

One process, several banks built one after the other (the previous one freed, a spacer allocation of a different size
kept alive so that the allocator hands out different addresses), the headline recall timed on each.  Prints the
dominant kernel's median time beside the addresses of the tensors it streams."""
import ctypes
import gc
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from aura_snn_rag_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rows, D, k, nq = 1_000_000, 768, 32, 2048
    spacers = []
    sizes = [0, 1 << 20, 3 << 20, (1 << 30) + (5 << 20), 7 << 16, (2 << 30) + 4096 * 37]
    for it, sp in enumerate(sizes):
        if sp:
            spacers.append(torch.empty(sp, dtype=torch.uint8, device=dev))
        hf = bench.new_bank(rows, D, dev)
        bench.fill_bank(hf, rows, D, 1234, dev)
        hf.rebuild_centroids(perm=torch.randperm(rows, generator=torch.Generator().manual_seed(7)))
        now = float(hf.memory_metadata[0, 1].item())
        g = torch.Generator(device=dev).manual_seed(99)
        pick = torch.randint(0, rows, (nq // 2,), generator=g, device=dev)
        q = torch.cat([hf.memory_features[pick] + 0.05 * torch.randn(nq // 2, D, generator=g, device=dev),
                       torch.randn(nq - nq // 2, D, generator=g, device=dev)]).contiguous()
        for _ in range(8):
            hf.recall_batch(q, k=k, now=now)
        torch.cuda.synchronize()
        ks = []
        flagsets = [int(x) for x in os.environ.get("AURA_PROBE_FLAGS", "0").split(",")]
        per_flag = {}
        for fl in flagsets:
            lib.aura_debug_cs_flags(fl)
            ks = []
            for rep in range(3):
                lib.aura_profile_begin(30 * 16)
                for _ in range(30):
                    hf.recall_batch(q, k=k, now=now)
                torch.cuda.synchronize()
                buf = (ctypes.c_float * (30 * 16))()
                n = lib.aura_profile_end(buf, 30 * 16)
                ks.append(sum(buf[j] for j in range(n)) / max(n, 1))
            per_flag[fl] = statistics.median(ks)
        lib.aura_debug_cs_flags(0)
        if len(flagsets) > 1:
            print(f"bank {it}: kernel ms per flag set " + ", ".join(f"{fl}: {v:.4f}" for fl, v in per_flag.items()), flush=True)
        ks = [per_flag[flagsets[0]]]
        iv = hf._ivf
        ptrs = {"sorted_bf16": iv.sorted_bf16.data_ptr(), "bank": hf.memory_features.data_ptr(),
                "rowc": iv.rowc.data_ptr() if getattr(iv, "rowc", None) is not None else 0}
        print(f"bank {it}: kernel {statistics.median(ks):.4f} ms; " +
              "; ".join(f"{n} {p:#x} (mod 2M {p % (2 << 20):#x}, mod 1G {p % (1 << 30):#x})" for n, p in ptrs.items()),
              flush=True)
        if os.environ.get("AURA_PROBE_REPACK"):          # same buffers, other list boundaries: does the mode follow?
            for seed in (8, 9, 10):
                hf.rebuild_centroids(perm=torch.randperm(rows, generator=torch.Generator().manual_seed(seed)))
                for _ in range(5):
                    hf.recall_batch(q, k=k, now=now)
                torch.cuda.synchronize()
                lib.aura_profile_begin(30 * 16)
                for _ in range(30):
                    hf.recall_batch(q, k=k, now=now)
                torch.cuda.synchronize()
                buf = (ctypes.c_float * (30 * 16))()
                n = lib.aura_profile_end(buf, 30 * 16)
                print(f"bank {it} after a rebuild with seed {seed}: kernel {sum(buf[j] for j in range(n)) / max(n, 1):.4f} ms; "
                      f"sorted_bf16 {hf._ivf.sorted_bf16.data_ptr():#x}", flush=True)
        if os.environ.get("AURA_PROBE_KEEP"):            # keep every bank alive: the next one lands on other pages
            spacers.append((hf, q, pick))
        else:
            del hf, q, pick
            gc.collect()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
