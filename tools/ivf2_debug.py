import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
from tests.test_gpu_knn import _clustered, _meta, _queries, _lists_of, NOW
dev = "cuda:0"
N, D, nq, k, ncent = [int(x) for x in sys.argv[1:6]]
g = torch.Generator().manual_seed(N + D + nq)
bank = _clustered(N, D, g, n_centres=300, spread=0.4).to(dev).contiguous()
inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
meta = _meta(N, g, decayed=True, spread_ts=True).to(dev).contiguous()
cent = torch.zeros(256, D, device=dev)
cent[:ncent] = bank[torch.randint(0, N, (ncent,), generator=g).to(dev)]
meta[:, 2] = ops.kmeans_assign(bank, cent, N, ncent).float()
meta[::97, 2] = -1.0
q = _queries(bank.cpu(), nq, g).to(dev).contiguous()
if ncent < 256: q[0] = 0.0
order, off, lens = _lists_of(meta, N)
srows, pad_off = ops.ivf2_layout(order, off, lens)
sshadow = ops.bank_shadow_sorted(bank, srows)
s1, r1, o1 = ops.knn_search_ivf2(bank, inv, meta, q, k, NOW, cent, 8, sshadow, srows, pad_off, lens)
s0, r0 = ops.knn_search(bank, inv, meta, q, k, NOW, centroids=cent, nprobe=8, fp32_scan=True)
cap = ops.ivf_capacity(int(torch.topk(lens, 8).values.sum().item()), k)
print("ovf", int(o1.item()), "cap", cap, "lens max", int(lens.max()), "min", int(lens.min()))
bad = (r0 != r1).any(1).nonzero().flatten()
print("bad", bad.numel(), "of", nq, bad[:10].tolist())
if cap is not None and nq <= 2048:
    s2, r2, o2 = ops.knn_search_ivf(bank, inv, meta, q, k, NOW, N, cent, 8, order, off, lens, cap)
    print("lists==masked", bool(torch.equal(r2, r0)), "lists==two-stage", bool(torch.equal(r2, r1)), "ovf lists", int(o2.item()))
for b in bad[:3].tolist():
    nv0, nv1 = int((r0[b] >= 0).sum()), int((r1[b] >= 0).sum())
    only1 = [r for r in r1[b].tolist() if r >= 0 and r not in r0[b].tolist()]
    only0 = [r for r in r0[b].tolist() if r >= 0 and r not in r1[b].tolist()]
    print("q", b, "valid masked", nv0, "two-stage", nv1, "only two-stage", only1[:6], [int(meta[r, 2]) for r in only1[:6]],
          "only masked", only0[:6], [int(meta[r, 2]) for r in only0[:6]])
    # probe set of this query per the masked scan: cids of its rows
    print("   cids masked rows", sorted(set(int(meta[r, 2]) for r in r0[b].tolist() if r >= 0))[:12],
          " two-stage rows", sorted(set(int(meta[r, 2]) for r in r1[b].tolist() if r >= 0))[:12])
print("---- masked fp32 path vs lists for several k")
for kk in (10, 33, 64, 100, 150, 200):
    capk = ops.ivf_capacity(int(torch.topk(lens, 8).values.sum().item()), kk)
    sl, rl, ol = ops.knn_search_ivf(bank, inv, meta, q, kk, NOW, N, cent, 8, order, off, lens, capk)
    for fd in (False, True):
        sm, rm = ops.knn_search(bank, inv, meta, q, kk, NOW, centroids=cent, nprobe=8, fp32_scan=True, force_dense=fd)
        print("k", kk, "force_dense", fd, "masked==lists", bool(torch.equal(rm, rl)), "valid masked", int((rm >= 0).sum()), "lists", int((rl >= 0).sum()))
print("---- arbitration for the first bad queries (torch reference over the rows of the listed cids)")
import torch.nn.functional as F
for b in bad[:4].tolist():
    cids_b = sorted(set(int(meta[r, 2]) for r in r0[b].tolist() if r >= 0) | set(int(meta[r, 2]) for r in r1[b].tolist() if r >= 0))
    rows = torch.nonzero(torch.isin(meta[:, 2].to(torch.int64), torch.tensor(cids_b, device=dev))).flatten()
    qn = F.normalize(q[b:b+1].double(), dim=1); bn = F.normalize(bank[rows].double(), dim=1)
    cos = (bn @ qn.T).flatten()
    sc = (0.5 * cos + 0.2 * torch.exp(-(NOW - meta[rows, 1].double()) / 3600.0)) * meta[rows, 0].double()
    top = torch.topk(sc, k)
    ref_rows = rows[top.indices]
    in0 = len(set(ref_rows.tolist()) & set(r0[b].tolist())); in1 = len(set(ref_rows.tolist()) & set(r1[b].tolist()))
    print("q", b, "cids", cids_b, "rows in those lists", rows.numel(), "ref top-k overlap: masked", in0, "two-stage", in1,
          "| best ref score", float(top.values[0]), "masked best", float(s0[b, 0]), "two-stage best", float(s1[b, 0]))
