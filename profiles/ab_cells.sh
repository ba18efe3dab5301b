#!/bin/bash
# A/B of cells_kernel builds (profiles/build_variants.sh): profiles/ab_cells.sh <variant> ...
mkdir -p gpurun_out; rm -f gpurun_out/ab_short.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/ab_tests.log
for v in "$@"; do
  TRX_LIB=$PWD/profiles/ab_libs/libtrx_$v.so timeout 300 python profiles/short_curves.py 100000 2>&1 | grep -E "^#|n_time +(50|100|200):" | sed "s/^/[$v] /" >> gpurun_out/ab_short.log
done
