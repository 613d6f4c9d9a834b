#!/bin/bash
mkdir -p gpurun_out/r03
(timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py -m gpu -x -q > gpurun_out/r03/t10.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t10.log; tail -3 gpurun_out/r03/t10.log | cut -c1-200)
bash tools/gpu_run10.sh
python tools/r03_shard_share.py 2>&1 | grep "^S="
