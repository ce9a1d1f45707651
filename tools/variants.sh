#!/bin/bash
# tools/variants.sh NAME "-DFLAG=.. -DFLAG=.." [NAME FLAGS]... — builds tuning variants of the library into .variants/
# (git-ignored, travels with gpurun); run one with CHUNKY_HIP_LIB=.variants/libchunky_hip_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p .variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  python3 - "$name" $flags <<'PY'
import sys
from chunkyclplugin_amd import native
name, flags = sys.argv[1], sys.argv[2:]
print("built", native.build(force=True, extra_flags=flags, out=f".variants/libchunky_hip_{name}.so", objdir=f".variants/obj_{name}"))
PY
done
