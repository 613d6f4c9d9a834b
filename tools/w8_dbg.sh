#!/bin/bash
# AURA_CS_DBG ablations of the 8-wave filter kernel on bench.py's workload
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for w in ${WAVES:-1}; do
for d in ${@:-0 32 34 35 39}; do
  if [ $w = 1 ]; then export AURA_CS_WAVES8=1; else unset AURA_CS_WAVES8; fi
  AURA_CS_DBG=$d timeout -k 10 200 python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 100 > $ROOT/gpurun_out/w8_dbg.json 2> $ROOT/gpurun_out/w8_dbg.err || { tail -5 $ROOT/gpurun_out/w8_dbg.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$ROOT/gpurun_out/w8_dbg.json'))
print('waves8=$w dbg=$d', 'ms/step', round(d['ms_per_step'],4), 'filter kernel_ms', round(d['roofline']['avg_kernel_ms'],4))
"
done
done
