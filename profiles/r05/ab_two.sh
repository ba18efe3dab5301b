#!/bin/bash
# bench.py (grid mode, no extras) of an A/B library against the tree's, alternating:  bash profiles/r05/ab_two.sh <variant> [runs]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
V=$1; N=${2:-2}
L=""
for k in $(seq 1 $N); do
TRX_LIB=$R/profiles/ab_libs/libtrx_$V.so python bench.py --no-cpu-baseline --no-e2e --no-batch-leg --pmc off > $O/ab_${V}_$k.json 2>/dev/null
python bench.py --no-cpu-baseline --no-e2e --no-batch-leg --pmc off > $O/ab_tree_$k.json 2>/dev/null
L="$L $O/ab_${V}_$k.json $O/ab_tree_$k.json"
done
python profiles/r05/ab_show.py $L
