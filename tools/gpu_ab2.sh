#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 500 python tools/ab_headline.py $1 --reps ${2:-7} > gpurun_out/r03/ab2.log 2>gpurun_out/r03/ab2.err || { tail -5 gpurun_out/r03/ab2.err; exit 1; }
cat gpurun_out/r03/ab2.log
