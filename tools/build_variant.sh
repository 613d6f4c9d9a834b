#!/bin/bash
# Builds aura_snn_rag_amd/lib/variants/libaura_<name>.so with extra -D flags for ONE translation unit:
#   tools/build_variant.sh pf6 aura_knn.hip -DAURA_CS_PF16=6
#   tools/build_variant.sh u8 aura_neuron.hip -DAURA_RTC_UNROLL=8
# Run a benchmark against it with AURA_HIP_LIB=<path of the .so>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
cd $ROOT/aura_snn_rag_amd/csrc
mkdir -p ../lib/variants /tmp/aura_variants
make -s aura_neuron.o aura_knn.o aura_bank.o aura_zone.o aura_train.o
objs=""
for o in aura_neuron aura_knn aura_bank aura_zone aura_train; do
  if [ "$o.hip" == "$src" ]; then
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src -o /tmp/aura_variants/${o}_$name.o
    objs="$objs /tmp/aura_variants/${o}_$name.o"
  else
    objs="$objs $o.o"
  fi
done
hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../lib/variants/libaura_$name.so
echo built $name
