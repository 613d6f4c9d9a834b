"""Randomised agreement sweep: two-stage recall (fp32 rows and bf16 shadow) vs the all-fp32 scan,
bit for bit, over random shapes / data / metadata (test infrastructure).
python tests/fuzz_two_stage.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops

def sweep(cases=40, seed=0, dev=None, verbose=True):
    """Returns the number of mismatching cases."""
    dev = torch.device("cuda:0") if dev is None else torch.device(dev)
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    bad = 0
    for c in range(cases):
        N = ri(8192, 140000)
        D = 4 * ri(1, 192)
        nq = [1, 3, 17, 64, 100, 256, 300, 777][ri(0, 7)]
        k = [1, 5, 10, 32, 64, 200][ri(0, 5)]
        kind = ri(0, 3)
        x = torch.randn(N, D, generator=g)
        if kind == 1:                                   # clustered
            cen = torch.randn(32, D, generator=g)
            x = cen[torch.randint(0, 32, (N,), generator=g)] + 0.3 * x
        elif kind == 2:                                 # wide norm range
            x = x * torch.exp(3 * torch.randn(N, 1, generator=g))
        elif kind == 3:                                 # many near-duplicates
            x[: N // 2] = x[torch.randint(0, 64, (N // 2,), generator=g)] + 1e-3 * torch.randn(N // 2, D, generator=g)
        bank = x.to(dev).contiguous()
        inv = torch.empty(N, device=dev); ops.bank_row_norms(bank, inv, 0, N)
        meta = torch.zeros(N, 4, device=dev)
        mk = ri(0, 2)
        meta[:, 0] = 1.0 if mk == 0 else (0.2 + 0.8 * torch.rand(N, generator=g)).to(dev)
        now = 1.7e9
        meta[:, 1] = now - (0.0 if mk == 0 else (7200.0 * torch.rand(N, generator=g)).to(dev))
        if mk == 2:                                     # fresh rows at the end of the bank (append order)
            meta[:, 1] = now - torch.linspace(7200.0, 0.0, N, device=dev)
        q = (bank[torch.randint(0, N, (nq,), generator=g).to(dev)] +
             0.3 * torch.randn(nq, D, generator=g).to(dev) * bank.std()).contiguous()
        s0, i0 = ops.knn_search(bank, inv, meta, q, k, now, fp32_scan=True)
        s1, i1 = ops.knn_search(bank, inv, meta, q, k, now)
        ok = torch.equal(i0, i1) and torch.equal(s0, s1)
        ovf = int(ops._overflow_flag(dev).item())
        txt = f"case {c}: N={N} D={D} nq={nq} k={k} kind={kind} meta={mk}: fp32-rows {'ok' if ok else 'MISMATCH'} (fallback={ovf})"
        if D % 8 == 0:
            sh, rho = ops.make_shadow(bank, inv)
            s2, i2 = ops.knn_search(bank, inv, meta, q, k, now, shadow=sh, rho=rho)
            ok2 = torch.equal(i0, i2) and torch.equal(s0, s2)
            txt += f" | shadow {'ok' if ok2 else 'MISMATCH'} (fallback={int(ops._overflow_flag(dev).item())})"
            ok = ok and ok2
            del sh
        if verbose or not ok:
            print(txt, flush=True)
        bad += 0 if ok else 1
        del bank, inv, meta, q
    return bad


if __name__ == "__main__":
    n_bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching cases:", n_bad)
    sys.exit(1 if n_bad else 0)
