#!/bin/bash
# tools/variants.sh NAME "-DFLAG=.. -DFLAG=.." [NAME FLAGS]... — builds tuning variants of the library into .variants/
# (git-ignored, travels with gpurun); run one with CHUNKY_HIP_LIB=.variants/libchunky_hip_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p .variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $flags \
      chunkyclplugin_amd/csrc/kernels.hip chunkyclplugin_amd/csrc/capi.hip chunkyclplugin_amd/csrc/widetree.cpp \
      -o .variants/libchunky_hip_$name.so && echo built $name ) &
done
wait
