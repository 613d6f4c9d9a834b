#!/bin/bash
# AURA_CS_DBG ablations on bench.py's workload (bits: 1 no MFMA loop, 2 no epilogue, 4 no prefetch)
cd /tmp && export TMPDIR=/tmp
for d in ${@:-0 1 2 3 7}; do
  AURA_CS_DBG=$d timeout -k 10 200 python3 /root/repo/bench.py --no-secondary --no-cpu-baseline --steps 100 > /root/repo/gpurun_out/bench_dbg.json 2> /root/repo/gpurun_out/bench_dbg.err || { tail -5 /root/repo/gpurun_out/bench_dbg.err; exit 1; }
  python3 -c "
import json; d=json.load(open('/root/repo/gpurun_out/bench_dbg.json'))
print('dbg=$d', 'ms/step', round(d['ms_per_step'],4), 'filter kernel_ms', round(d['roofline']['avg_kernel_ms'],4))
"
done
