"""Dev probe: two-stage (bf16 prefilter + fp32 re-score) recall vs the fp32 scan: equality + time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops

def bank_of(N, D, dev, seed=0, clustered=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N, D, generator=g)
    if clustered:
        c = torch.randn(64, D, generator=g)
        x = c[torch.randint(0, 64, (N,), generator=g)] + 0.3 * x
    bank = x.to(dev)
    inv = torch.empty(N, device=dev)
    ops.bank_row_norms(bank, inv, 0, N)
    meta = torch.zeros(N, 4, device=dev)
    meta[:, 0] = 1.0 - 0.3 * torch.rand(N, generator=g).to(dev)
    meta[:, 1] = 1.7e9 - 5000 * torch.rand(N, generator=g).to(dev)
    return bank, inv, meta

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n

def main():
    dev = torch.device("cuda:0")
    cases = [(100000, 768, 256, 32, False), (100000, 768, 256, 32, True), (50000, 512, 100, 10, False),
             (20011, 200, 37, 5, False), (100000, 768, 1, 5, False), (300000, 256, 600, 64, True)]
    if len(sys.argv) > 1:
        cases = cases[: int(sys.argv[1])]
    for N, D, nq, k, cl in cases:
        bank, inv, meta = bank_of(N, D, dev, clustered=cl)
        g = torch.Generator().manual_seed(1)
        q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.5 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
        now = 1.7e9 + 100.0
        s0, i0 = ops.knn_search(bank, inv, meta, q, k, now, fp32_scan=True)
        s1, i1 = ops.knn_search(bank, inv, meta, q, k, now, check_overflow=False)
        torch.cuda.synchronize()
        same_i = bool((i0 == i1).all()); same_s = bool((s0 == s1).all())
        ovf = int(ops._ovf_flags[dev].item())
        sh_txt = ""
        if D % 8 == 0:
            shadow = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
            ops.bank_shadow_update(bank, shadow)
            assert torch.equal(shadow, bank.to(torch.bfloat16))
            s2, i2 = ops.knn_search(bank, inv, meta, q, k, now, check_overflow=False, shadow=shadow)
            ok2 = bool((i0 == i2).all()) and bool((s0 == s2).all())
            ovf2 = int(ops._ovf_flags[dev].item())
            t_s = timed(lambda: ops.knn_search(bank, inv, meta, q, k, now, check_overflow=False, shadow=shadow))
            sh_txt = f" | shadow: equal={ok2} overflow={ovf2} {t_s*1e3:.1f}us"
        t_f = timed(lambda: ops.knn_search(bank, inv, meta, q, k, now, fp32_scan=True, check_overflow=False))
        t_c = timed(lambda: ops.knn_search(bank, inv, meta, q, k, now, check_overflow=False))
        print(f"N={N} D={D} nq={nq} k={k} clustered={cl}: idx_equal={same_i} score_equal={same_s} overflow={ovf} "
              f"fp32={t_f*1e3:.1f}us coarse={t_c*1e3:.1f}us  maxdiff={(s0-s1).abs().max().item():.3e}" + sh_txt, flush=True)
        if not same_i:
            bad = (i0 != i1).any(1).nonzero().flatten()[:3]
            for b in bad.tolist():
                print("  q", b, i0[b].tolist()[:8], i1[b].tolist()[:8], s0[b].tolist()[:4], s1[b].tolist()[:4])
        del bank, inv, meta

if __name__ == "__main__":
    main()
