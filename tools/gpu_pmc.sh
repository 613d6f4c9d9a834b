#!/bin/bash
# PMC passes (gpurun_out/pmc_r03 -> profiles/r03_pmc_per_dispatch.json travels back through gpurun_out/)
mkdir -p gpurun_out/r03
bash tools/pmc_collect.sh 2>&1 | tail -25
cp profiles/r03_pmc_per_dispatch.json gpurun_out/r03/r03_pmc_per_dispatch.json
