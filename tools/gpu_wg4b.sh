#!/bin/bash
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  if [ $v == 1 ]; then export AURA_IVF_WG4=1; else unset AURA_IVF_WG4; fi
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/pg$v -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/bg$v.json 2> $R/gpurun_out/r03/bg$v.err || exit 1
  echo "== WG4=$v"; python3 $R/tools/kstats.py $R/gpurun_out/r03/pg$v/p_kernel_stats.csv 10 | grep "coarse\|probe\|plan\|slots\|thresh"
  python3 $R/tools/bench_summary.py $R/gpurun_out/r03/bg$v.json | head -1
done
