"""Where does the bf16 GIF loop spend its time?  Same neuron-steps with and without the streams."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aura_snn_rag_amd import ops
dev = torch.device("cuda:0")
rows, T, H = 8192, 16, 3072
for dt in (torch.bfloat16, torch.float32):
    h3 = (torch.randn(rows, T, H, device=dev) * 2).to(dt); h2 = h3[:, 0].contiguous()
    o3 = torch.empty_like(h3); o2 = torch.empty_like(h2)
    v = torch.zeros(rows, H, device=dev, dtype=dt); th = torch.ones_like(v)
    for ti, mo in ((False, False), (True, False), (False, True), (True, True)):
        ms = bench.timed_events(lambda: ops.gif_run(h2 if ti else h3, o2 if mo else o3, v, th, math.exp(-0.1), 8, 0.01, 1.0, T,
                                                     time_invariant=ti, mean_out=mo))
        print(dt, "time_invariant", ti, "mean_out", mo, f"{ms:.4f} ms  {rows*T*H/ms/1e6:.1f} G steps/s", flush=True)
