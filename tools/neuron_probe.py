"""Runs each fused neuron kernel a few times (for rocprofv3 --pmc / --stats)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n
if which in ("all", "izh_nt"):
    for N, T in ((1 << 22, 100), (1 << 20, 128), (1 << 20, 100)):
        I = 20 * torch.rand(N, T, device=dev); S = torch.empty_like(I)
        v = torch.full((N,), -65.0, device=dev); u = 0.2 * v
        ms = timed(lambda: ops.izh_run_nt(I, S, v, u, 0.02, 0.2, -65.0, 8.0, 0.2))
        print(f"izh_nt N={N} T={T}: {ms:.3f} ms  {N*T/ms/1e6:.1f} Gsteps/s  {(N*T*8+N*16)/ms/1e6:.0f} GB/s")
        del I, S, v, u
if which in ("all", "gif"):
    for dt in (torch.bfloat16, torch.float32):
        rows, T, H = 8192, 16, 3072
        h = (torch.randn(rows, T, H, device=dev) * 2).to(dt); out = torch.empty_like(h)
        vv = torch.zeros(rows, H, device=dev, dtype=dt); th = torch.ones_like(vv)
        ms = timed(lambda: ops.gif_run(h, out, vv, th, math.exp(-0.1), 8, 0.01, 1.0, T))
        b = h.element_size()
        print(f"gif {dt} rows={rows} T={T} H={H}: {ms:.3f} ms  {rows*T*H/ms/1e6:.1f} Gsteps/s  {(rows*T*H*2*b+rows*H*4*b)/ms/1e6:.0f} GB/s")
        del h, out, vv, th
if which in ("all", "btd"):
    B, T, D = 4096, 100, 1024
    I = 20 * torch.rand(B, T, D, device=dev); S = torch.empty_like(I)
    v = torch.full((B * D,), -65.0, device=dev); u = 0.2 * v
    ms = timed(lambda: ops.izh_run_btd(I, S, v, u, 0.02, 0.2, -65.0, 8.0, 0.2))
    print(f"izh_btd: {ms:.3f} ms  {B*D*T/ms/1e6:.1f} Gsteps/s  {(B*D*T*8+B*D*16)/ms/1e6:.0f} GB/s")
