#!/bin/bash
# LDS / issue counters of the two-stage filter kernel (separate --pmc passes, kernel-trace only).
# usage: bash tools/pmc_lds.sh [tag]   (honours AURA_CS_DBG / AURA_CS_WAVES8 from the environment)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
OUT=$ROOT/gpurun_out/pmc_lds_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
pass() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
pass lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pass lds2 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
pass iss1 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES
pass iss2 SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass iss3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM
python3 $ROOT/tools/pmc_summarize.py $OUT $ROOT/gpurun_out/pmc_lds_$TAG.json | grep "coarse_scan_kernel<24, 1"
find $OUT -name "*.csv" -size +2M -delete
