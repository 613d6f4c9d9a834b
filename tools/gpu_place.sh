#!/bin/bash
mkdir -p gpurun_out/r03
: > gpurun_out/r03/place.log
AURA_PROBE_REPACK=1 AURA_PROBE_KEEP=1 timeout -k 10 700 python tools/placement_probe.py 2>&1 | grep "^bank\|Error\|error" | cut -c1-110 >> gpurun_out/r03/place.log
cat gpurun_out/r03/place.log
