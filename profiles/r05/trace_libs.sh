#!/bin/bash
# mean duration of the chain's kernels on ONE stream (16 TOIs, two steps) for several libraries ("tree" = the tree's):
#   bash profiles/r05/trace_libs.sh tree variant1 variant2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=""
for V in "$@"; do
  if [ "$V" = tree ]; then ARGS="$ARGS TRX_TAG=tree"; else ARGS="$ARGS TRX_LIB=$R/profiles/ab_libs/libtrx_$V.so"; fi
done
bash $R/profiles/r05/trace_env.sh $ARGS
