#!/bin/bash
# A/B of library builds on the short-curve throughput table: profiles/ab_short.sh <rows> "<n_time ...>" lib1 lib2 ...
# (lib = a file under profiles/ab_libs/ or "default")
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=$1; NT=$2; shift 2
for L in "$@"; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$R/profiles/ab_libs/libtrx_$L.so; fi
  for rep in 1 2; do python $R/profiles/short_curves.py $ROWS $NT 2>&1 | grep -v amdgpu.ids | sed "s/^/[$L] /"; done
done
