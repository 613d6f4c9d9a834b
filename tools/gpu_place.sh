#!/bin/bash
mkdir -p gpurun_out/r03
: > gpurun_out/r03/place.log
for rep in 1 2; do AURA_PROBE_FLAGS="0,262144" AURA_PROBE_KEEP=1 timeout -k 10 600 python tools/placement_probe.py 2>&1 | grep "per flag set\|Error\|error" | cut -c1-100 >> gpurun_out/r03/place.log; echo "--" >> gpurun_out/r03/place.log; done
cat gpurun_out/r03/place.log
