#!/bin/bash
# timing ablations of the inverted-list filter launch at the headline (results are wrong with these bits)
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 2048 32; do
  export AURA_CS_DBG=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/pa$v -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 5 > /dev/null 2> $R/gpurun_out/r03/pa$v.err
  echo "== AURA_CS_DBG=$v"; python3 $R/tools/kstats.py $R/gpurun_out/r03/pa$v/p_kernel_stats.csv 6 | grep "coarse_scan"
done
