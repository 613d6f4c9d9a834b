#!/bin/bash
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default 0; do
  if [ $v == 0 ]; then export AURA_RF_WAVE=0; fi
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/pw$v -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/bw$v.json 2> $R/gpurun_out/r03/bw$v.err || exit 1
  echo "== AURA_RF_WAVE=$v"; python3 $R/tools/kstats.py $R/gpurun_out/r03/pw$v/p_kernel_stats.csv 8 | grep "refine\|coarse_scan"
  python3 $R/tools/bench_summary.py $R/gpurun_out/r03/bw$v.json | head -1
done
