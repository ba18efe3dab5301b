#!/bin/bash
# A/B builds of libtrx.so under profiles/ab_libs/ (git-ignored, shipped to the GPU box):
#   profiles/build_variants.sh name1 "-DFLAG=.. -DFLAG2=.." name2 "..." ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/profiles/ab_libs
while [ $# -gt 1 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm $FLAGS \
      -o $R/profiles/ab_libs/libtrx_$NAME.so $R/triceratops_amd/csrc/trx_kernels.hip $R/triceratops_amd/csrc/trx_draw.hip $R/triceratops_amd/csrc/trx_scenario.hip 2>&1 | grep -E "error|Error" 
  echo "built $NAME ($FLAGS)"
done
