#!/bin/bash
# Hardware counters per dispatch for bench.py's recall step, in separate rocprofv3 --pmc passes
# (never combined with tracing domains other than --kernel-trace).  Run on the GPU box from the
# repo root:  bash tools/pmc_collect.sh   -> gpurun_out/pmc_r01/*.csv + profiles/r01_pmc_per_dispatch.json
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_r01
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2"
pass() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
# FETCH_SIZE (3 TCC counters) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md)
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass busy GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
pass valu SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
python3 $ROOT/tools/pmc_summarize.py $OUT $ROOT/profiles/r01_pmc_per_dispatch.json
# keep only the summary-sized files in gpurun_out
find $OUT -name "*.csv" -size +8M -delete
