#!/bin/bash
mkdir -p gpurun_out/r03
AURA_BENCH_PARITY_DEBUG=1 timeout -k 10 300 python bench.py --no-secondary --steps 20 --warmup 5 > gpurun_out/r03/pd_plain.json 2> gpurun_out/r03/pd_plain.err
python -c "import json; d=json.load(open('gpurun_out/r03/pd_plain.json')); print('no-secondary:', d['cpu_baseline']['gpu_parity_on_sample'])"
grep "parity" gpurun_out/r03/pd_plain.err | head -12
AURA_BENCH_PARITY_DEBUG=1 timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/pd_full.json 2> gpurun_out/r03/pd_full.err
python -c "import json; d=json.load(open('gpurun_out/r03/pd_full.json')); print('with secondary:', d['cpu_baseline']['gpu_parity_on_sample'])"
grep "parity" gpurun_out/r03/pd_full.err | head -30
