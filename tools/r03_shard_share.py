"""Timing probe (one GPU): ONE rank's share of the row-sharded layouts of BASELINE config 4 -- 1M/S rows, S x 2048
all-gathered queries, local recall only -- with each shard's own sampled bound (round 2) and with the exchanged bound
(round 3).  The exchange is EMULATED: the other shards' bounds are not available on one GPU, so the combined bound
is taken as max(own k-th, own ceil(k/S)-th) -- what min-over-shards of the ceil(k/S)-th bounds is for statistically
identical shards.  Timing only: results under the emulated bound are not the recall's (a real exchange is tested in
tests/test_gpu_scale.py and tests/test_gpu_sharded_r03.py).  python tools/r03_shard_share.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import new_bank, fill_bank, timed_wall

dev = torch.device("cuda")
D, k = 768, 32
for S in (1, 2, 8):
    rows = 1_000_000 // S
    hf = new_bank(rows, D, dev)
    fill_bank(hf, rows, D, 1234, dev)
    torch.manual_seed(7)
    hf.rebuild_centroids()
    now = float(hf.memory_metadata[0, 1].item())
    g = torch.Generator(device=dev).manual_seed(99)
    nq = 2048 * S
    q = torch.randn(nq, D, generator=g, device=dev)
    ids = hf.probe(q)
    t_own = timed_wall(lambda: hf.recall_batch(q, k=k, now=now, probe_ids=ids, fallback_empty=False), 20, warm=5)
    line = f"S={S}: {rows} rows x {nq} queries: own bound {t_own * 1e3:.3f} ms"
    if S > 1:
        def fn(b):
            return torch.maximum(b[:, 0], b[:, 1]).contiguous()
        t_ex = timed_wall(lambda: hf.recall_batch(q, k=k, now=now, probe_ids=ids, fallback_empty=False,
                                                  bound_exchange=(fn, S)), 20, warm=5)
        line += f", exchanged bound (emulated) {t_ex * 1e3:.3f} ms -> {2048 * S / t_ex:.3e} retrievals/s per step of all ranks"
    print(line, flush=True)
    del hf
    torch.cuda.empty_cache()
