#!/bin/bash
# kernel resource usage of one .hip file: tools/kres.sh aura_knn.hip <grep pattern>
cd "$(dirname "$0")/../aura_snn_rag_amd/csrc" || exit 1
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 --offload-device-only -c "$1" -o /tmp/kres_dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/kres_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kres_dev.elf || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kres_dev.elf | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|agpr_count" | paste - - - - - | grep "${2:-.}" | sed 's/  */ /g' | cut -c1-260
