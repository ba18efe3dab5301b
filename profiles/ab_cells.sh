#!/bin/bash
# A/B of cells_kernel builds (profiles/build_variants.sh): profiles/ab_cells.sh <variant> ...
mkdir -p gpurun_out; rm -f gpurun_out/ab_short.log
for v in "$@"; do
  TRX_LIB=$PWD/profiles/ab_libs/libtrx_$v.so timeout 300 python profiles/short_curves.py 100000 $AB_TIMES 2>&1 | grep -E "n_time" | sed "s/^/[$v] /" >> gpurun_out/ab_short.log
done
cat gpurun_out/ab_short.log
