#!/bin/bash
# AURA_RF_DBG ablations of coarse_refine_kernel on bench.py's workload (1: stop after the candidate load,
# 2: stop after select + compaction, 4: skip the re-scoring loop)
cd /tmp && export TMPDIR=/tmp
for d in ${@:-0 1 2 4}; do
  rm -rf /root/repo/gpurun_out/prof_rf
  AURA_RF_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_rf -o b -- python3 /root/repo/bench.py --no-secondary --no-cpu-baseline --steps 30 > /dev/null 2> /root/repo/gpurun_out/rf_dbg.err || { tail -5 /root/repo/gpurun_out/rf_dbg.err; exit 1; }
  echo "rf_dbg=$d $(python3 /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_rf/b_results.db coarse_refine | tail -1)"
done
rm -rf /root/repo/gpurun_out/prof_rf
