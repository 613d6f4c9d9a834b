#!/bin/bash
mkdir -p gpurun_out/r03
(timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py tests/test_gpu_online_write.py -m gpu -x -q > gpurun_out/r03/t8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t8.log; tail -5 gpurun_out/r03/t8.log | cut -c1-300)
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | tee gpurun_out/r03/b6_plain.json | cut -c1-200
timeout -k 10 200 python tools/r03_write_probe.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r03/w6.log
