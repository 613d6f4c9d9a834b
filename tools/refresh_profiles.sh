#!/bin/bash
# Regenerates profiles/r01_bench.json and profiles/r01_bench_kernel_stats.csv on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_r01
timeout -k 10 500 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_r01 -o b -- python3 $ROOT/bench.py > $ROOT/gpurun_out/bench_r01.json 2> $ROOT/gpurun_out/bench_r01.err || { tail -20 $ROOT/gpurun_out/bench_r01.err; exit 1; }
python3 $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_r01/b_results.db > $ROOT/profiles/r01_bench_kernel_stats.csv
cp $ROOT/gpurun_out/bench_r01.json $ROOT/profiles/r01_bench.json
cp $ROOT/profiles/r01_bench.json $ROOT/profiles/r01_bench_kernel_stats.csv $ROOT/gpurun_out/
rm -rf $ROOT/gpurun_out/prof_r01
head -12 $ROOT/profiles/r01_bench_kernel_stats.csv
