#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_r03.py tests/test_gpu_scale.py tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py -m gpu -x -q > gpurun_out/r03/t20.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r03/t20.log; tail -3 gpurun_out/r03/t20.log | cut -c1-200
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED\|error" gpurun_out/r03/t20.log | head -20; exit 1; }
timeout -k 10 300 python tools/r03_shard_share.py 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 tools/bench_summary.py /dev/stdin | head -1
