#!/bin/bash
# GPU call: whole GPU suite + write probe + headline profile + plain bench
mkdir -p gpurun_out/r03
(timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t4.log; tail -25 gpurun_out/r03/t4.log)
timeout -k 10 200 python tools/r03_write_probe.py 2>&1 | tee gpurun_out/r03/w4.log || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p4 -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/b4.json 2> $R/gpurun_out/r03/b4.err || exit 1
cd $R && python tools/kstats.py gpurun_out/r03/p4/p_kernel_stats.csv 8
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 | tee gpurun_out/r03/b4_plain.json | cut -c1-300
