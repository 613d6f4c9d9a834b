#!/bin/bash
mkdir -p gpurun_out/r03
: > gpurun_out/r03/place.log
for rep in 1 2; do
 echo "== contiguous" >> gpurun_out/r03/place.log
 AURA_PROBE_KEEP=1 timeout -k 10 500 python tools/placement_probe.py 2>&1 | grep "^bank\|Error\|error" | cut -c1-60 >> gpurun_out/r03/place.log
 echo "== ordinary" >> gpurun_out/r03/place.log
 AURA_NO_CONTIGUOUS=1 AURA_PROBE_KEEP=1 timeout -k 10 500 python tools/placement_probe.py 2>&1 | grep "^bank\|Error\|error" | cut -c1-60 >> gpurun_out/r03/place.log
done
cat gpurun_out/r03/place.log
