#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_training_bf16.py tests/test_gpu_training.py tests/test_gpu_neurons.py tests/test_golden.py -m gpu -q > gpurun_out/r03/t13.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/t13.log; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" gpurun_out/r03/t13.log | cut -c1-260 | tail -40
