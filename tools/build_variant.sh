#!/bin/bash
# Builds aura_snn_rag_amd/lib/variants/libaura_<name>.so with extra -D flags for aura_knn.hip:
#   tools/build_variant.sh pf6 -DAURA_CS_PF16=6
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $ROOT/aura_snn_rag_amd/csrc
mkdir -p ../lib/variants /tmp/aura_variants
make -s aura_neuron.o aura_bank.o aura_zone.o aura_train.o
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c aura_knn.hip -o /tmp/aura_variants/knn_$name.o
hipcc -shared -fPIC --offload-arch=gfx950 aura_neuron.o aura_bank.o /tmp/aura_variants/knn_$name.o aura_zone.o aura_train.o -o ../lib/variants/libaura_$name.so
echo built $name
