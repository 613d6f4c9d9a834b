#!/bin/bash
# per-kernel times of the two-stage recall (args: number of probe cases)
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_coarse
timeout -k 10 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_coarse -o cp -- python3 /root/repo/tools/coarse_probe.py ${1:-2} > /root/repo/gpurun_out/coarse_prof.log 2>&1 || { tail -20 /root/repo/gpurun_out/coarse_prof.log; exit 1; }
grep "N=" /root/repo/gpurun_out/coarse_prof.log
python3 /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_coarse/cp_results.db GLOBAL
rm -rf /root/repo/gpurun_out/prof_coarse
