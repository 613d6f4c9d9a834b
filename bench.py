#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): retrievals/s of the episodic cosine-kNN.  One "step" = one 256-query
batch per rank recalled against the whole bank (top-32), inputs resident in HBM.

  N = 1  : BASELINE config 2 -- 100 000 x 768 fp32 bank, 256-query batch, top-32.
  N > 1  : the same bank row-sharded over N ranks (100 000 / N rows each); every rank brings its
           own 256-query batch; a step = all-gather the queries (RCCL), scan the local shard for
           all N*256 queries, all-gather the per-shard top-k, merge.  Per-GPU scan work is
           constant in N ("weak"): value = N*256*steps / time.  --bank-rows 1000000 gives the
           1 M-row bank of config 4.

Also reported on the same JSON line: `roofline` of the dominant kernel (main MFMA scan, timed
with HIP events on its launch stream via aura_profile_*), `cpu_baseline` (the oracle's
reference-cost recall on the host cores, rank 0, N = 1 only) and `secondary` (neuron-timestep
throughput of the fused Izhikevich / GIF loops with their HBM roofline fractions).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3    # fp32 matrix peak (v_mfma_f32_32x32x2_f32), dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--bank-rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=24)
    return ap.parse_args()


def make_shard(rows, dim, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    # generate in chunks to bound host memory
    out = torch.empty(rows, dim, device=dev)
    step = 65536
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        out[r0:r0 + n] = torch.randn(n, dim, generator=g).to(dev)
    return out


def secondary_neurons(dev):
    """Fused neuron-loop throughput: Izhikevich 2^22 x 100 (time-contiguous layout) and the GIF
    loop of one SNNFFN layer at config 3 (512 x 16 x 3072 bf16)."""
    from aura_snn_rag_amd import ops
    res = {}

    def timed(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        st = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        en = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        for i in range(iters):
            st[i].record(); fn(); en[i].record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in zip(st, en))
        return ms[len(ms) // 2]

    N, T = 1 << 22, 100
    I = 20 * torch.rand(N, T, device=dev)
    S = torch.empty_like(I)
    v = torch.full((N,), -65.0, device=dev); u = 0.2 * v
    ms = timed(lambda: ops.izh_run_nt(I, S, v, u, 0.02, 0.2, -65.0, 8.0, 0.2))
    bytes_alg = N * T * 8 + N * 16
    res["izhikevich_nt"] = {"neurons": N, "timesteps": T, "dtype": "f32",
                            "neuron_timesteps_per_s": N * T / (ms * 1e-3),
                            "ms": ms, "algorithmic_bytes": bytes_alg,
                            "hbm_gbs": bytes_alg / (ms * 1e-3) / 1e9,
                            "hbm_frac_of_8TBs": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del I, S, v, u
    B, T2, D = 4096, 100, 1024
    I = 20 * torch.rand(B, T2, D, device=dev); S = torch.empty_like(I)
    v = torch.full((B * D,), -65.0, device=dev); u = 0.2 * v
    ms = timed(lambda: ops.izh_run_btd(I, S, v, u, 0.02, 0.2, -65.0, 8.0, 0.2))
    bytes_alg = B * D * T2 * 8 + B * D * 16
    res["izhikevich_btd"] = {"neurons": B * D, "timesteps": T2, "dtype": "f32",
                             "neuron_timesteps_per_s": B * D * T2 / (ms * 1e-3), "ms": ms,
                             "algorithmic_bytes": bytes_alg,
                             "hbm_gbs": bytes_alg / (ms * 1e-3) / 1e9,
                             "hbm_frac_of_8TBs": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del I, S, v, u
    # GIF at config 3, interface traffic: h in + spikes out + state, bf16; many rows to fill HBM
    rows, T3, H = 8192, 16, 3072
    h = (torch.randn(rows, T3, H, device=dev) * 2).to(torch.bfloat16)
    out = torch.empty_like(h)
    vv = torch.zeros(rows, H, device=dev, dtype=torch.bfloat16); th = torch.ones_like(vv)
    import math
    ms = timed(lambda: ops.gif_run(h, out, vv, th, math.exp(-0.1), 8, 0.01, 1.0, T3))
    bytes_alg = rows * T3 * H * 4 + rows * H * 8
    res["gif_bf16"] = {"rows": rows, "timesteps": T3, "hidden": H, "dtype": "bf16",
                       "neuron_timesteps_per_s": rows * T3 * H / (ms * 1e-3), "ms": ms,
                       "algorithmic_bytes": bytes_alg,
                       "hbm_gbs": bytes_alg / (ms * 1e-3) / 1e9,
                       "hbm_frac_of_8TBs": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del h, out, vv, th
    # whole spiking FFN at config 3 (d=768, H=3072, S=512, T=16, L=8, bf16): T-deduplicated GEMMs
    # (vendor library) + the two fused GIF loops; neuron-steps = rows * T * (H + D)
    from aura_snn_rag_amd.core.language_zone.snn_ffn import SNNFFN
    torch.manual_seed(0)
    ffn = SNNFFN(768, 3072, num_timesteps=16, L=8).to(dev).to(torch.bfloat16).eval()
    x = torch.randn(1, 512, 768, device=dev, dtype=torch.bfloat16)
    def ffn_forward():
        with torch.no_grad():
            return ffn(x)
    ms = timed(ffn_forward)
    res["snnffn_config3_bf16"] = {"ms": ms, "neuron_timesteps_per_s": 512 * 16 * (3072 + 768) / (ms * 1e-3),
                                  "note": "module forward incl. 4 GEMMs; reference CPU path measured 315 ms (BASELINE.md)"}
    # one-shot write throughput (config 5 seeding path): rows of 768 fp32 into the bank
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    hf = HippocampalFormation(feature_dim=768, max_memories=1 << 20, n_place_cells=8, n_time_cells=4,
                              n_grid_cells=4, device="cuda", use_centroid_index=False)
    rows_w = 1 << 18
    fw = torch.randn(rows_w, 768, device=dev)
    def wr():
        hf.memory_count = 0
        hf.bulk_write(fw, rebuild=False)
    ms = timed(wr, iters=5)
    res["bulk_write"] = {"rows": rows_w, "dim": 768, "ms": ms, "rows_per_s": rows_w / (ms * 1e-3),
                         "hbm_gbs": rows_w * 768 * 8 / (ms * 1e-3) / 1e9}
    return res


def centroid_index_recall(dev, bank, inv, meta, q, k, now, steps=30, shadow=None):
    """The reference's use_centroid_index retrieval (8 nearest of 256 centroids per query,
    hippocampal.py:259-270) through the inverted-list kernels: same bank and queries as the
    headline run, index = one Lloyd iteration from 256 random rows (rebuild_centroids)."""
    from aura_snn_rag_amd import _lib, ops
    lib = _lib.load()
    N, D = bank.shape
    g = torch.Generator(device="cpu").manual_seed(7)
    cent = torch.zeros(256, D, device=dev)
    cent[:] = bank[torch.randperm(N, generator=g)[:256].to(dev)]
    assign = ops.kmeans_assign(bank, cent, N, 256)
    ops.kmeans_update(bank, assign, cent, 256, update_means=True)
    assign = ops.kmeans_assign(bank, cent, N, 256)
    meta_i = meta.clone()
    meta_i[:, 2] = assign.float()
    cids = assign
    order = torch.sort(cids, stable=True).indices.to(torch.int32).contiguous()
    lens = torch.bincount(cids.long(), minlength=256)[:256].to(torch.int32).contiguous()
    off = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev),
                     torch.cumsum(lens, 0).to(torch.int32)]).contiguous()

    cap = ops.ivf_capacity(int(torch.topk(lens, 8).values.sum().item()), k)

    def step():
        return ops.knn_search_ivf(bank, inv, meta_i, q, k, now, N, cent, 8, order, off, lens, cap)
    s_i, r_i, ovf = step()
    # agreement with the masked full scan (same candidate sets): must be identical
    s_m, r_m = ops.knn_search(bank, inv, meta_i, q, k, now, count=N, centroids=cent, nprobe=8)
    same = bool(torch.equal(r_i, r_m)) and bool(torch.equal(s_i, s_m)) and int(ovf.item()) == 0
    # recall@k of the pruned search against the exact search
    s_e, r_e = ops.knn_search(bank, inv, meta_i, q, k, now, count=N)
    hit = (r_i.unsqueeze(2) == r_e.unsqueeze(1)).any(dim=2).float().mean().item()
    top1 = (r_i[:, 0] == r_e[:, 0]).float().mean().item()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib.aura_profile_begin(steps * 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buf = (ctypes.c_float * (steps * 2))()
    n = lib.aura_profile_end(buf, steps * 2)
    kms = sum(buf[i] for i in range(n)) / max(n, 1)
    probed_rows = int(lens.sum().item())      # at nq = 256 every list is probed by some query
    bytes_alg = probed_rows * (D * 4 + 24) + q.shape[0] * D * 4
    # the same candidate restriction applied inside the two-stage scan (probe masks in LDS, bf16 shadow
    # rows): what HippocampalFormation uses for banks up to a few hundred thousand rows
    masked = None
    if shadow is not None:
        def mstep():
            return ops.knn_search(bank, inv, meta_i, q, k, now, count=N, centroids=cent, nprobe=8,
                                  shadow=shadow, check_overflow=False)
        s_t, r_t = mstep()
        same_t = bool(torch.equal(r_t, r_i)) and bool(torch.equal(s_t, s_i))
        for _ in range(3):
            mstep()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            mstep()
        torch.cuda.synchronize()
        dm = time.perf_counter() - t1
        masked = {"retrievals_per_s": q.shape[0] * steps / dm, "ms_per_step": dm / steps * 1e3,
                  "identical_to_inverted_lists": same_t}
    return {"retrievals_per_s": q.shape[0] * steps / dt, "ms_per_step": dt / steps * 1e3,
            "two_stage_masked": masked,
            "nprobe": 8, "lists": 256, "identical_to_masked_full_scan": same,
            "recall_at_k_vs_exact": hit, "top1_agreement_vs_exact": top1,
            "roofline": {"bound": "hbm", "kernel": "ivf_scan_kernel", "achieved": bytes_alg / (kms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "avg_kernel_ms": kms, "algorithmic_bytes_per_launch": bytes_alg, "traffic": None}}


def centroid_index_1m(dev, k=32, N=1_000_000, D=768, nq=2048, steps=10):
    """north_star target: >= 1e6 retrievals/s at a 1M x 768 bank on one MI355X.  The reference's
    centroid-index retrieval (8 nearest of 256 centroids) on a 1M-row bank, 2048-query batches: the
    inverted lists on the two-stage scan (list-sorted bf16 shadow) next to the fp32 lists; identical
    results.  HBM roofline of the prefilter launch: every list is streamed once per batch."""
    from aura_snn_rag_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(11)
    bank = torch.empty(N, D, device=dev)
    for r0 in range(0, N, 1 << 17):
        bank[r0:r0 + (1 << 17)] = torch.randn(min(1 << 17, N - r0), D, generator=g).to(dev)
    inv = torch.empty(N, device=dev)
    ops.bank_row_norms(bank, inv, 0, N)
    now = 1.7e9
    meta = torch.zeros(N, 4, device=dev)
    meta[:, 0] = 1.0; meta[:, 1] = now
    cent = bank[torch.randperm(N, generator=g)[:256].to(dev)].clone()
    assign = ops.kmeans_assign(bank, cent, N, 256)
    ops.kmeans_update(bank, assign, cent, 256, update_means=True)
    assign = ops.kmeans_assign(bank, cent, N, 256)
    meta[:, 2] = assign.float()
    order = torch.sort(assign, stable=True).indices.to(torch.int32).contiguous()
    lens = torch.bincount(assign.long(), minlength=256)[:256].to(torch.int32).contiguous()
    off = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(lens, 0).to(torch.int32)]).contiguous()
    cap = ops.ivf_capacity(int(torch.topk(lens, 8).values.sum().item()), k)
    srows, pad_off = ops.ivf2_layout(order, off, lens)
    sshadow = ops.bank_shadow_sorted(bank, srows)
    q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.5 * torch.randn(nq, D, generator=g).to(dev)).contiguous()

    def two_stage():
        return ops.knn_search_ivf2(bank, inv, meta, q, k, now, cent, 8, sshadow, srows, pad_off, lens)

    def lists_fp32():
        return ops.knn_search_ivf(bank, inv, meta, q, k, now, N, cent, 8, order, off, lens, cap)
    s1, r1, o1 = two_stage()
    flag = int(o1.item())
    s0, r0, _ = lists_fp32()
    same = bool(torch.equal(r0, r1)) and bool(torch.equal(s0, s1)) and flag == 0
    out = {"bank_rows": N, "dim": D, "queries_per_batch": nq, "k": k, "nprobe": 8, "lists": 256,
           "identical_results": same}
    for name, fn in (("two_stage_lists", two_stage), ("fp32_lists", lists_fp32)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        lib.aura_profile_begin(steps * 2)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        buf = (ctypes.c_float * (steps * 2))()
        n = lib.aura_profile_end(buf, steps * 2)
        e = {"retrievals_per_s": nq * steps / dt, "ms_per_batch": dt / steps * 1e3}
        if name == "two_stage_lists" and n > 0:
            kms = sum(buf[i] for i in range(n)) / n
            nbytes = int(srows.numel()) * (D * 2 + 16)
            e["roofline"] = {"bound": "hbm", "kernel": "coarse_scan_kernel<24,FILTER,bf16 rows,IVF>",
                             "achieved": nbytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_kernel_ms": kms,
                             "algorithmic_bytes_per_launch": nbytes, "traffic": None}
        out[name] = e
    del bank, sshadow
    return out


def cpu_baseline(bank_rows, dim, k, nq_sample):
    """The oracle's recall at the REFERENCE's cost model (bank re-normalised per query,
    hippocampal.py:273-279) on the host cores; bit-identical to the reference (tests)."""
    from oracle import aura_oracle as O
    # the GPU box gives a 1-GPU job a 16-core CPU share (all 256 host cores are visible, but
    # oversubscribing them is slower than using the share)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    g = torch.Generator().manual_seed(1234)
    ob = O.OracleBank(bank_rows, dim, use_centroid_index=False)
    ob.features = torch.randn(bank_rows, dim, generator=g)
    ob.metadata[:, 0] = 1.0
    now = 1.7e9
    ob.metadata[:, 1] = now
    ob.count = bank_rows
    q = torch.randn(nq_sample, dim, generator=g)
    ob.recall(q[0], k, now)  # warm-up
    t0 = time.perf_counter()
    for i in range(nq_sample):
        ob.recall(q[i], k, now)
    dt = time.perf_counter() - t0
    return {"value": nq_sample / dt, "unit": "retrievals/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{nq_sample} single-query recalls (reference algorithm: per-query bank "
                      f"normalise + mm + topk) over the same {bank_rows}x{dim} fp32 bank, k={k}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("AURA_BENCH_FORCE_DIST") == "1"   # 1-rank RCCL smoke test
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from aura_snn_rag_amd import _lib, ops
    from aura_snn_rag_amd.sharded import ShardedRecall, shard_rows
    lib = _lib.load()

    D, k, nq = args.dim, args.k, args.nq
    r0, r1 = shard_rows(args.bank_rows, world, rank)
    rows = r1 - r0
    bank = make_shard(rows, D, 1234 + rank, dev)
    inv = torch.empty(rows, device=dev)
    ops.bank_row_norms(bank, inv, 0, rows)
    now = 1.7e9
    meta = torch.zeros(rows, 4, device=dev)
    meta[:, 0] = 1.0; meta[:, 1] = now; meta[:, 2] = -1
    g = torch.Generator(device="cpu").manual_seed(99 + rank)
    pick = torch.randint(0, rows, (nq // 2,), generator=g)
    q = torch.cat([bank[pick.to(dev)] + 0.05 * torch.randn(nq // 2, D, generator=g).to(dev),
                   torch.randn(nq - nq // 2, D, generator=g).to(dev)]).contiguous()

    # bf16 shadow of the rows, kept beside the bank like 1/||row|| (HippocampalFormation maintains
    # both on every write): the two-stage recall's prefilter streams it instead of the fp32 rows
    shadow = None
    if D % 8 == 0 and D <= 768 and rows >= 8192:
        shadow = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
        ops.bank_shadow_update(bank, shadow)

    def local_search(qq, kk, check=False, fp32_scan=False, use_shadow=True):
        return ops.knn_search(bank, inv, meta, qq, kk, now, count=rows, idx_base=r0, check_overflow=check,
                              fp32_scan=fp32_scan, shadow=shadow if use_shadow else None)

    recall = ShardedRecall(local_search, ops.topk_merge)

    def step(check=False):
        if not use_dist:
            return local_search(q, k, check)
        return recall.recall(q, k, all_gather_queries=True)

    # correctness guard before timing: overflow check + planted neighbours found
    s, i = step(check=True) if not use_dist else step()
    torch.cuda.synchronize()
    planted_ok = bool((i[: nq // 2, 0].cpu() == (pick + r0).to(torch.int32)).float().mean() > 0.99)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    lib.aura_profile_begin(max(1, args.steps * 4))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    buf = (ctypes.c_float * (args.steps * 4))()
    nprof = lib.aura_profile_end(buf, args.steps * 4)
    if use_dist:
        t = torch.tensor([elapsed], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_q = nq * world * args.steps
    value = total_q / elapsed
    # dominant kernel of the default (two-stage) path: the bf16 prefilter scan, which reads the bank
    # once -> HBM roofline; algorithmic bytes per launch = rows * (D*4 + 16 B of row constants).
    # (Shards below 8192 rows, D > 768 etc. run the fp32 matrix scan instead: MFMA roofline.)
    def roofline_of(buf_ms, n, steps):
        if n <= 0:
            return None
        avg_ms = sum(buf_ms[j] for j in range(n)) / n
        rows_c, nq_c = ctypes.c_int64(0), ctypes.c_int64(0)
        lib.aura_profile_last_scan(ctypes.byref(rows_c), ctypes.byref(nq_c))
        scanned_rows, nq_launch = rows_c.value, nq_c.value
        flop = 2.0 * nq_launch * scanned_rows * D
        tf = flop / (avg_ms * 1e-3) / 1e12
        common = {"traffic": None, "avg_kernel_ms": avg_ms, "launches_timed": n,
                  "rows_per_launch": scanned_rows, "queries_per_launch": nq_launch,
                  "launches_per_step": n / steps}
        kind = lib.aura_profile_last_scan_kind()
        if kind in (1, 2):
            esz = 2 if kind == 2 else 4                      # bf16 shadow rows / fp32 rows
            nbytes = scanned_rows * (D * esz + 16)
            gbs = nbytes / (avg_ms * 1e-3) / 1e9
            return dict(common, bound="hbm",
                        kernel=f"coarse_scan_kernel<KS,FILTER,{'bf16 shadow' if kind == 2 else 'fp32'} rows> "
                               "(rows streamed once by global_load_lds, v_mfma_f32_16x16x32_bf16 against "
                               "register-resident query fragments)",
                        achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                        algorithmic_bytes_per_launch=nbytes,
                        pmc_key=("coarse_scan_kernel<24, 1, true, false, false, 8>" if kind == 2
                                 else "coarse_scan_kernel<24, 1, false, false, false, 4>"),
                        bf16_tflops_at_kernel=tf, bf16_frac_of_2500=tf / 2500.0)
        return dict(common, bound="mfma",
                    kernel="knn_scan_filter_v2 (v_mfma_f32_32x32x2_f32, fp32 in / fp32 acc)",
                    achieved=tf, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=tf / MFMA_F32_PEAK_TF,
                    algorithmic_flop_per_launch=flop, pmc_key="knn_scan_filter_v2<true, true>",
                    algorithmic_bytes_per_launch=scanned_rows * (D * 4 + 24) + nq_launch * D * 4,
                    hbm_gbs_at_kernel=(scanned_rows * D * 4) / (avg_ms * 1e-3) / 1e9)

    def add_traffic(roof):
        # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (PMC and timing
        # runs must not be mixed); the committed per-dispatch summary is reported with the gfx950
        # correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide streaming read)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_per_dispatch.json")
        if roof is None or not os.path.exists(pmc) or args.bank_rows != 100_000 or world != 1:
            return
        try:
            d = json.load(open(pmc)).get(roof["pmc_key"], {})
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                roof["traffic"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
                roof["traffic_source"] = ("profiles/r01_pmc_per_dispatch.json (rocprofv3 --pmc FETCH_SIZE / "
                                          "WRITE_SIZE, bytes per launch)")
        except Exception:
            pass

    roof = roofline_of(buf, nprof, args.steps)
    add_traffic(roof)

    # the all-fp32 scan of the same workload (AURA_KNN_FP32_SCAN): same results bit for bit, bound by
    # the fp32 matrix pipe; kept as a second measured line
    fp32_line = None
    if rank == 0 and world == 1 and roof is not None and roof["bound"] == "hbm":
        for _ in range(3):
            local_search(q, k, fp32_scan=True)
        torch.cuda.synchronize()
        lib.aura_profile_begin(max(1, args.steps * 4))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            local_search(q, k, fp32_scan=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        buf2 = (ctypes.c_float * (args.steps * 4))()
        n2 = lib.aura_profile_end(buf2, args.steps * 4)
        r2 = roofline_of(buf2, n2, args.steps)
        add_traffic(r2)
        fp32_line = {"retrievals_per_s": nq * args.steps / el, "ms_per_step": el / args.steps * 1e3, "roofline": r2}
        if shadow is not None:                               # two-stage path streaming the fp32 rows
            for _ in range(3):
                local_search(q, k, use_shadow=False)
            torch.cuda.synchronize()
            lib.aura_profile_begin(max(1, args.steps * 4))
            t1 = time.perf_counter()
            for _ in range(args.steps):
                local_search(q, k, use_shadow=False)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            n3 = lib.aura_profile_end(buf2, args.steps * 4)
            r3 = roofline_of(buf2, n3, args.steps)
            add_traffic(r3)
            fp32_line["two_stage_without_shadow"] = {"retrievals_per_s": nq * args.steps / el,
                                                     "ms_per_step": el / args.steps * 1e3, "roofline": r3}

    out = {
        "metric": "retrievals/sec", "value": value, "unit": "retrievals/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"episodic cosine-kNN recall: {args.bank_rows}x{D} fp32 bank "
                               f"({'row-sharded over %d ranks' % world if world > 1 else 'one GPU'}), "
                               f"{nq}-query batch per rank, top-{k}, exact fp32 scores (bf16 matrix-core "
                               f"prefilter over a bf16 shadow of the rows with a proven error bound + fp32 "
                               f"re-scoring of the survivors from the fp32 bank: rows and score bits "
                               f"identical to the all-fp32 scan)",
                   "bank_rows": args.bank_rows, "dim": D, "queries_per_rank": nq, "k": k,
                   "parallelism": f"bank-sharded x{world}" if world > 1 else "single"},
        "planted_neighbours_found": planted_ok,
        "roofline": roof,
    }
    if fp32_line is not None:
        out["fp32_scan_only"] = fp32_line
    if rank == 0 and world == 1 and not args.no_secondary:
        out["secondary"] = secondary_neurons(dev)
        out["secondary"]["centroid_index_recall"] = centroid_index_recall(dev, bank, inv, meta, q, k, now,
                                                                          shadow=shadow)
        del bank, shadow
        torch.cuda.empty_cache()
        out["secondary"]["centroid_index_1m"] = centroid_index_1m(dev, k=k)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.bank_rows, D, k, args.cpu_queries)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
