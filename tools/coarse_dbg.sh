#!/bin/bash
# timing ablations of the bf16 prefilter scan (AURA_CS_DBG bits: 1 no MFMA loop, 2 no epilogue, 4 no prefetch)
cd /tmp && export TMPDIR=/tmp
for d in ${@:-0 1 2 3 4 7}; do
  rm -rf /root/repo/gpurun_out/prof_dbg
  AURA_CS_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_dbg -o cp -- python3 /root/repo/tools/coarse_probe.py 1 > /root/repo/gpurun_out/dbg$d.log 2>&1 || exit 1
  echo "dbg=$d"; python3 /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_dbg/cp_results.db coarse_scan
done
rm -rf /root/repo/gpurun_out/prof_dbg
