#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_online_write.py tests/test_gpu_bank_r02.py tests/test_gpu_ingest_r02.py -m gpu -x -q > gpurun_out/r03/t_onl.log 2>&1; rc=$?; tail -3 gpurun_out/r03/t_onl.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
echo "== v2"; timeout -k 10 200 python tools/r03_write_probe.py 2>&1 | grep -v amdgpu | tail -8
echo "== v1"; AURA_ONLINE_V1=1 timeout -k 10 200 python tools/r03_write_probe.py 2>&1 | grep -v amdgpu | tail -8
