#!/bin/bash
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/ponl -o p -- python3 $R/tools/r03_write_probe.py > $R/gpurun_out/r03/ponl.log 2>&1
cd $R; python3 tools/kstats.py gpurun_out/r03/ponl/p_kernel_stats.csv 14 | cut -c1-150
