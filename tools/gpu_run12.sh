#!/bin/bash
mkdir -p gpurun_out/r03
(timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py tests/test_gpu_neurons.py -m gpu -x -q > gpurun_out/r03/t12.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t12.log; tail -3 gpurun_out/r03/t12.log | cut -c1-200)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p12 -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/b12.json 2> $R/gpurun_out/r03/b12.err || exit 1
cd $R; python3 tools/kstats.py gpurun_out/r03/p12/p_kernel_stats.csv 6 | grep "coarse"
python3 tools/bench_summary.py gpurun_out/r03/b12.json | head -1
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 30 | cut -c1-120
