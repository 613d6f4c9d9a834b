#!/bin/bash
# Hardware counters per dispatch for bench.py's recall steps and the neuron loops, in separate rocprofv3 --pmc
# passes (never combined with tracing domains other than --kernel-trace).  Run on the GPU box from the repo root:
#   bash tools/pmc_collect.sh   -> gpurun_out/pmc_r03/<config>/<pass>/... + profiles/r03_pmc_per_dispatch.json
# AURA_PMC_ONLY="cfg1 cfg2" restricts the configs (a full run is ~20 rocprofv3 launches).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_r03
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-secondary --steps 6 --warmup 2"
want() { [ -z "$AURA_PMC_ONLY" ] || [[ " $AURA_PMC_ONLY " == *" $1 "* ]]; }
pass() {  # config, name, program + flags, counters...
  local cfg=$1 name=$2 prog=$3; shift 3
  mkdir -p $OUT/$cfg
  rm -rf $OUT/$cfg/$name
  echo "[pmc] $cfg $name" >&2
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$cfg/$name -o p -- python3 $prog > $OUT/$cfg/$name.log 2>&1 || { tail -5 $OUT/$cfg/$name.log; return 1; }
}
config() {  # config key, program + flags
  want "$1" || return 0
  # FETCH_SIZE (3 TCC counters) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md)
  pass "$1" fetch "$2" FETCH_SIZE
  pass "$1" write "$2" WRITE_SIZE
  pass "$1" busy "$2" GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
  pass "$1" valu "$2" SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
}
config headline_1000000x768_n1_index "$ROOT/bench.py $COMMON"
config exact_1000000x768 "$ROOT/bench.py $COMMON --exact --nq 256"
config exact_1000000x768_2048q "$ROOT/bench.py $COMMON --exact --nq 2048"
config config2 "$ROOT/bench.py $COMMON --bank-rows 100000 --nq 256 --exact"
config neurons "$ROOT/tools/neuron_bench.py"
python3 $ROOT/tools/pmc_summarize.py $OUT $ROOT/profiles/r03_pmc_per_dispatch.json
# keep only the summary-sized files in gpurun_out
find $OUT -name "*.csv" -size +8M -delete
