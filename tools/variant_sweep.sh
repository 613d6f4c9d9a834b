#!/bin/bash
# Runs bench.py's headline workload against every library under aura_snn_rag_amd/lib/variants/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/aura_snn_rag_amd/lib/variants/libaura_*.so; do
  n=$(basename $so .so)
  AURA_HIP_LIB=$so timeout -k 10 200 python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps ${STEPS:-200} > $ROOT/gpurun_out/variant_$n.json 2> $ROOT/gpurun_out/variant_$n.err || { tail -5 $ROOT/gpurun_out/variant_$n.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$ROOT/gpurun_out/variant_$n.json'))
print('$n', 'ms/step', round(d['ms_per_step'],4), 'filter kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))
"
done
