#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py -m gpu -x -q > gpurun_out/r03/t15.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r03/t15.log; tail -3 gpurun_out/r03/t15.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
bash tools/gpu_ab2.sh "$1" ${2:-5}
