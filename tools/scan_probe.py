"""Times the main scan kernel (aura_profile hooks) for several feature dims: separates the
k-loop cost from the fixed prologue/epilogue cost."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
N, nq, k = 100_000, 256, 32
for D in (32, 64, 128, 256, 768, 1536):
    bank = torch.randn(N, D, device=dev); inv = torch.empty(N, device=dev)
    ops.bank_row_norms(bank, inv, 0, N)
    meta = torch.zeros(N, 4, device=dev); meta[:, 0] = 1; meta[:, 1] = 1.7e9; meta[:, 2] = -1
    q = torch.randn(nq, D, device=dev)
    for _ in range(3): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False)
    torch.cuda.synchronize()
    lib.aura_profile_begin(64)
    for _ in range(10): ops.knn_search(bank, inv, meta, q, k, 1.7e9, check_overflow=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 64)(); n = lib.aura_profile_end(buf, 64)
    ms = sorted(buf[i] for i in range(n))[n // 2]
    print(f"D={D:5d} main-scan {ms*1e3:8.1f} us   k-tiles={D//32}  per-ktile {ms*1e3/(D//32):6.2f} us")
