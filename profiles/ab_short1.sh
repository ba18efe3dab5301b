#!/bin/bash
# like ab_short.sh, one repetition, batches only: profiles/ab_short1.sh <rows> "<n_time ...>" lib1 lib2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=$1; NT=$2; shift 2
for L in "$@"; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$R/profiles/ab_libs/libtrx_$L.so; fi
  python $R/profiles/short_curves.py $ROWS $NT $JITTER 2>&1 | grep n_time | sed "s/^/[$L] /"
done
