#!/bin/bash
# functional rehearsal of the N > 1 bench path on one GPU: RCCL at world size 1 with forced collectives, then two
# ranks sharing the card over gloo (AURA_BENCH_BACKEND=gloo)
mkdir -p gpurun_out/r03
AURA_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03/dist1.json 2> gpurun_out/r03/dist1.err; echo "rccl world-1 rc $?"
python3 tools/bench_summary.py gpurun_out/r03/dist1.json | head -1
AURA_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r03/dist2.json 2> gpurun_out/r03/dist2.err; echo "gloo 2 ranks rc $?"
tail -1 gpurun_out/r03/dist2.json | python3 -c "import sys, json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','n_gpus','ms_per_step','scaling','collective_backend')}, d['config']['parallelism'])"
