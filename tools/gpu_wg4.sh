#!/bin/bash
mkdir -p gpurun_out/r03
export AURA_IVF_WG4=1
(timeout -k 10 500 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py -m gpu -x -q > gpurun_out/r03/t_wg4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t_wg4.log; tail -6 gpurun_out/r03/t_wg4.log | cut -c1-300)
if grep -q "pytest rc 0" gpurun_out/r03/t_wg4.log; then
  for i in 1 2; do
    AURA_IVF_WG4=1 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
    env -u AURA_IVF_WG4 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
  done
fi
