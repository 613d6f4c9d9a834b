#!/bin/bash
# third GPU call of round 3: whole GPU suite (new sharded tests included) + headline profile
mkdir -p gpurun_out/r03
(timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t3.log; tail -25 gpurun_out/r03/t3.log)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p3 -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/b3.json 2> $R/gpurun_out/r03/b3.err || exit 1
cd $R && python tools/kstats.py gpurun_out/r03/p3/p_kernel_stats.csv 8
