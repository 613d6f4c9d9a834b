#!/bin/bash
mkdir -p gpurun_out/r03
for rep in 1 2; do timeout -k 10 400 python tools/placement_probe.py 2>&1 | grep "^bank" >> gpurun_out/r03/place.log; echo "--" >> gpurun_out/r03/place.log; done
cut -c1-330 gpurun_out/r03/place.log
