#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_shard
timeout -k 10 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_shard -o s -- python3 /root/repo/tools/shard_probe.py > /root/repo/gpurun_out/shard_probe.log 2>&1 || { tail /root/repo/gpurun_out/shard_probe.log; exit 1; }
grep world /root/repo/gpurun_out/shard_probe.log
python3 /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_shard/s_results.db GLOBAL | head -12
rm -rf /root/repo/gpurun_out/prof_shard
