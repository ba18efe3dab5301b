#!/bin/bash
# 64-TOI steps and 75-scenario calc_probs of several builds of the library, alternating, in one job:
#   bash profiles/r06/ab_steps.sh <tag> libA.so tree ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
T=$1; shift
for rep in 1 2; do
for L in "$@"; do
  P=$R/profiles/ab_libs/$L; [ "$L" = tree ] && P=$R/triceratops_amd/libtrx.so
  echo "== $L rep $rep"
  TRX_LIB=$P python profiles/r05/batch_step.py 5 2>/dev/null | grep "step [2345]" | cut -c1-18 | tr '\n' ' '; echo
  TRX_LIB=$P python profiles/r05/e2e_step.py 4 2>/dev/null | grep "run [1234]" | tr '\n' ' '; echo
done
done | tee $O/ab_steps_${T}.txt
