#!/bin/bash
# 64-TOI steps and 75-scenario calc_probs of an A/B library against the tree's, alternating:  bash profiles/r05/ab_steps.sh <variant>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$1
for k in 1 2; do
echo "== $V"; TRX_LIB=$R/profiles/ab_libs/libtrx_$V.so python profiles/r05/batch_step.py 4 2>/dev/null | grep "step [234]" | cut -c1-18 | tr '\n' ' '; echo
TRX_LIB=$R/profiles/ab_libs/libtrx_$V.so python profiles/r05/e2e_step.py 3 2>/dev/null | grep "run [123]" | tr '\n' ' '; echo
echo "== tree"; python profiles/r05/batch_step.py 4 2>/dev/null | grep "step [234]" | cut -c1-18 | tr '\n' ' '; echo
python profiles/r05/e2e_step.py 3 2>/dev/null | grep "run [123]" | tr '\n' ' '; echo
done
