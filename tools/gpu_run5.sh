#!/bin/bash
# quick GPU call: recall-related tests + headline profile + plain bench
mkdir -p gpurun_out/r03
(timeout -k 10 500 python -m pytest tests/test_gpu_knn.py tests/test_gpu_knn_r03.py tests/test_gpu_scale.py tests/test_gpu_sharded_r03.py tests/test_gpu_bank_r02.py -m gpu -x -q > gpurun_out/r03/t7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t7.log; tail -6 gpurun_out/r03/t7.log)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p5 -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/b5.json 2> $R/gpurun_out/r03/b5.err || exit 1
cd $R && python tools/kstats.py gpurun_out/r03/p5/p_kernel_stats.csv 10
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 | tee gpurun_out/r03/b5_plain.json | cut -c1-300
