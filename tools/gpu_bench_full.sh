#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 1000 python bench.py > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err; echo "bench rc $?"
tail -3 gpurun_out/r03/bench_full.err
python tools/bench_summary.py gpurun_out/r03/bench_full.json
