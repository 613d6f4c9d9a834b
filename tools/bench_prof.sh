#!/bin/bash
# quick headline measurement + per-kernel times of bench.py's recall step (no secondary, no CPU baseline)
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_bench -o b -- python3 /root/repo/bench.py --no-secondary --no-cpu-baseline --steps 100 > /root/repo/gpurun_out/bench_quick.json 2> /root/repo/gpurun_out/bench_quick.err || { tail -20 /root/repo/gpurun_out/bench_quick.err; exit 1; }
python3 -c "
import json; d=json.load(open('/root/repo/gpurun_out/bench_quick.json'))
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'roof', d['roofline']['bound'], round(d['roofline']['frac'],3), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))
print('fp32 only', round(d['fp32_scan_only']['retrievals_per_s']), d['fp32_scan_only']['roofline']['frac'])
"
python3 /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_bench/b_results.db GLOBAL
rm -rf /root/repo/gpurun_out/prof_bench
