#!/bin/bash
# kernel durations of one pass of the 18 families on short curves: profiles/stats_cells.sh <tag> <n_time> <rows> [lib]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; NT=$2; NR=$3
[ -n "$4" ] && export TRX_LIB=$R/profiles/ab_libs/libtrx_$4.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG} -- python3 $R/profiles/cells_once.py $NT $NR cells > $R/gpurun_out/stats_${TAG}.log 2>&1
F=$(ls $R/gpurun_out/stats_${TAG}/*/*kernel_stats.csv | head -1)
echo "== $TAG"; cut -d, -f1-4 $F | sed 's/(anonymous namespace):://' | cut -c1-110 | head -8
