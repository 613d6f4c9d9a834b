#!/bin/bash
mkdir -p gpurun_out/r03
(timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py -m gpu -x -q > gpurun_out/r03/t9.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t9.log; tail -4 gpurun_out/r03/t9.log | cut -c1-200)
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
  AURA_RF_WAVE=0 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
done
python tools/r03_shard_share.py 2>&1 | grep "^S="
