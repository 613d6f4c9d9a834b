#!/bin/bash
# headline step time vs the sample size of the two-stage recall (AURA_CS_SAMPLE_ROWS)
cd /tmp && export TMPDIR=/tmp
for r in ${@:-4096 6144 8192 12288 16384 24576}; do
  AURA_CS_SAMPLE_ROWS=$r timeout -k 10 200 python3 /root/repo/bench.py --no-secondary --no-cpu-baseline --steps 300 > /root/repo/gpurun_out/ss.json 2> /root/repo/gpurun_out/ss.err || { tail -3 /root/repo/gpurun_out/ss.err; exit 1; }
  python3 -c "
import json; d=json.load(open('/root/repo/gpurun_out/ss.json'))
print('sample_rows=$r', 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value']), 'filter_ms', round(d['roofline']['avg_kernel_ms'],4))
"
done
