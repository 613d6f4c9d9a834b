"""Dev probe: inverted-list recall through the two-stage scan vs the fp32 lists path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0"); k = int(os.environ.get("K", "32"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nqs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 2048]
g = torch.Generator().manual_seed(0)
bank = torch.empty(N, D, device=dev)
for r0 in range(0, N, 1 << 17):
    bank[r0:r0 + (1 << 17)] = torch.randn(min(1 << 17, N - r0), D, generator=g).to(dev)
inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9
cent = bank[torch.randperm(N, generator=g)[:256].to(dev)].clone()
assign = ops.kmeans_assign(bank, cent, N, 256); ops.kmeans_update(bank, assign, cent, 256)
assign = ops.kmeans_assign(bank, cent, N, 256); meta[:, 2] = assign.float()
meta[::1001, 2] = -1
cids = meta[:, 2].to(torch.int32)
order = torch.sort(cids, stable=True).indices.to(torch.int32).contiguous()
valid = cids >= 0
lens = torch.bincount(cids.clamp(min=0).long(), weights=valid.float(), minlength=256)[:256].to(torch.int32).contiguous()
n_neg = (N - valid.sum()).to(torch.int32).reshape(1)
off = torch.cat([n_neg, n_neg + torch.cumsum(lens, 0).to(torch.int32)]).contiguous()
cap = ops.ivf_capacity(int(torch.topk(lens, 8).values.sum().item()), k)
srows, pad_off = ops.ivf2_layout(order, off, lens)
sshadow = ops.bank_shadow_sorted(bank, srows)
print(f"N={N} D={D} lists min {int(lens.min())} max {int(lens.max())} n_sorted {srows.numel()} cap {cap}")
for nq in nqs:
    q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.5 * torch.randn(nq, D, generator=g).to(dev)).contiguous()
    s0, r0_, o0 = ops.knn_search_ivf(bank, inv, meta, q, k, 1.7e9, N, cent, 8, order, off, lens, cap)
    s1, r1, o1 = ops.knn_search_ivf2(bank, inv, meta, q, k, 1.7e9, cent, 8, sshadow, srows, pad_off, lens)
    torch.cuda.synchronize()
    same = bool(torch.equal(r0_, r1) and torch.equal(s0, s1))
    print(f"  nq={nq}: identical={same} overflow(lists)={int(o0.item())} overflow(two-stage)={int(o1.item())}", flush=True)
    if not same:
        bad = (r0_ != r1).any(1).nonzero().flatten()
        print("   bad queries", bad[:10].tolist(), "of", bad.numel())
        b = int(bad[0]); print("   ", r0_[b].tolist()[:8], r1[b].tolist()[:8], s0[b].tolist()[:4], s1[b].tolist()[:4])
        sm, rm = ops.knn_search(bank, inv, meta, q, k, 1.7e9, centroids=cent, nprobe=8, fp32_scan=True)
        print("   masked==lists", bool(torch.equal(rm, r0_)), "masked==two-stage", bool(torch.equal(rm, r1)))
        extra = [r for r in r1[b].tolist() if r >= 0 and r not in rm[b].tolist()]
        print("   rows only in two-stage:", extra[:8], "their cids", [int(meta[r, 2]) for r in extra[:8]])
        print("   n valid: lists", int((r0_[b] >= 0).sum()), "two-stage", int((r1[b] >= 0).sum()))
    for fn, name in ((lambda: ops.knn_search_ivf(bank, inv, meta, q, k, 1.7e9, N, cent, 8, order, off, lens, cap), "lists fp32"),
                     (lambda: ops.knn_search_ivf2(bank, inv, meta, q, k, 1.7e9, cent, 8, sshadow, srows, pad_off, lens), "two-stage")):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"    {name}: {dt*1e3:.3f} ms/batch -> {nq/dt:,.0f} retrievals/s")
