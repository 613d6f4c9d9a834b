"""Randomised agreement sweep: inverted lists on the two-stage scan and the masked two-stage scan vs the
masked fp32 scan, bit for bit (test infrastructure).
python tests/fuzz_ivf2.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops


def sweep(cases=30, seed=0, dev=None, verbose=True):
    dev = torch.device("cuda:0") if dev is None else torch.device(dev)
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    bad = 0
    for c in range(cases):
        N = ri(8192, 120000); D = 8 * ri(1, 96); nq = [1, 9, 100, 256, 700, 2500, 8300][ri(0, 6)]
        k = [1, 5, 32, 64, 150][ri(0, 4)]; ncent = [256, 256, 200, 60][ri(0, 3)]
        x = torch.randn(N, D, generator=g)
        if ri(0, 1):
            cen = torch.randn(64, D, generator=g); x = cen[torch.randint(0, 64, (N,), generator=g)] + 0.4 * x
        bank = x.to(dev).contiguous()
        inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
        meta = torch.zeros(N, 4, device=dev)
        meta[:, 0] = (0.3 + 0.7 * torch.rand(N, generator=g)).to(dev)
        now = 1.7e9
        meta[:, 1] = now - torch.linspace(7200.0, 0.0, N, device=dev) if ri(0, 1) else now - (7200 * torch.rand(N, generator=g)).to(dev)
        cent = torch.zeros(256, D, device=dev)
        cent[:ncent] = bank[torch.randint(0, N, (ncent,), generator=g).to(dev)]
        meta[:, 2] = ops.kmeans_assign(bank, cent, N, ncent).float()
        if ri(0, 1): meta[::53, 2] = -1.0
        q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] + 0.3 * bank.std() * torch.randn(nq, D, generator=g).to(dev)).contiguous()
        slack = [0, 64][ri(0, 1)]                      # packed lists / lists with free entries behind them
        st = ops.build_ivf2(bank, inv, meta[:, 2], slack=slack)
        s1, r1, o1 = ops.knn_search_ivf2(bank, inv, meta, q, k, now, cent, 8, st["sorted_bf16"], st["rho"],
                                         st["sorted_rows"], st["pad_off"], st["list_len"], n_sorted=st["n_sorted"])
        flag = int(o1.item()) & ~ops.KNN_FLAG_NO_CANDIDATES
        s0, r0 = ops.knn_search(bank, inv, meta, q, k, now, centroids=cent, nprobe=8, fp32_scan=True)
        # third path: the masked two-stage scan over the (unsorted) bf16 shadow
        shadow, rho = ops.make_shadow(bank, inv)
        s2, r2 = ops.knn_search(bank, inv, meta, q, k, now, centroids=cent, nprobe=8, shadow=shadow, rho=rho)
        masked_ok = bool(torch.equal(r0, r2) and torch.equal(s0, s2))
        del shadow
        if flag:
            okq = (r0 == r1).all(1)
            ok = bool(torch.equal(s0[okq], s1[okq]))        # overflowed queries may differ, the rest must not
            note = f"overflow flag {flag}, {int((~okq).sum())} queries affected"
        else:
            ok = bool(torch.equal(r0, r1) and torch.equal(s0, s1)); note = ""
        if not masked_ok: note += " | masked two-stage MISMATCH"
        ok = ok and masked_ok
        if verbose or not ok:
            print(f"case {c}: N={N} D={D} nq={nq} k={k} ncent={ncent}: {'ok' if ok else 'MISMATCH'} {note}", flush=True)
        bad += 0 if ok else 1
        del bank, st
    return bad


if __name__ == "__main__":
    n_bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching cases:", n_bad)
    sys.exit(1 if n_bad else 0)
