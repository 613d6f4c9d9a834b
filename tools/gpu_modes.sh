#!/bin/bash
mkdir -p gpurun_out/r03
L=gpurun_out/r03/modes.log; : > $L
for rep in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python tools/ab_headline.py 0 --reps 3 2>/dev/null | tr '\n' ' ' | sed 's/dominant kernel/k/g; s/step median/step/g; s/shader clock (s_memtime \/ s_memrealtime over 2 ms, after the runs)/clk/' >> $L; echo >> $L
done
cut -c1-250 $L
