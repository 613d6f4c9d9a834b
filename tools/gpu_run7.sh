#!/bin/bash
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
  AURA_RF_WAVE=0 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p7 -o p -- python3 $R/tools/r03_s8_profile.py > /dev/null 2> $R/gpurun_out/r03/p7.err
cd $R && python tools/kstats.py gpurun_out/r03/p7/p_kernel_stats.csv 12
