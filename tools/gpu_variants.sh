#!/bin/bash
# same-box comparison of library builds: tools/gpu_variants.sh "<name> <name> ..." (name "main" = the in-tree library)
mkdir -p gpurun_out/r03
L=gpurun_out/r03/variants.log; : > $L
for rep in 1 2 3; do for v in $1; do
  if [ $v = main ]; then unset AURA_HIP_LIB; else export AURA_HIP_LIB=$PWD/aura_snn_rag_amd/lib/variants/libaura_$v.so; fi
  echo -n "$v rep $rep: " >> $L
  timeout -k 10 200 python tools/ab_headline.py ${2:-0} --reps 3 2>/dev/null | tr '\n' ' ' | sed 's/dominant kernel/k/g; s/step median/step/g' >> $L; echo >> $L
done; done
cut -c1-300 $L
