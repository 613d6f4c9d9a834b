#!/bin/bash
# second GPU call of round 3: online-write tests, write probe, headline profile + plain bench
mkdir -p gpurun_out/r03
(timeout -k 10 300 python -m pytest tests/test_gpu_online_write.py tests/test_gpu_knn.py tests/test_gpu_bank_r02.py -m gpu -x -q > gpurun_out/r03/t2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t2.log; tail -15 gpurun_out/r03/t2.log)
timeout -k 10 200 python tools/r03_write_probe.py 2>&1 | tee gpurun_out/r03/w1.log || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/p2 -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r03/b2.json 2> $R/gpurun_out/r03/b2.err || exit 1
cd $R && python tools/kstats.py gpurun_out/r03/p2/p_kernel_stats.csv 12
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 | tee gpurun_out/r03/b2_plain.json | cut -c1-300
