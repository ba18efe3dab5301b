#!/bin/bash
# 64-TOI steps and 75-scenario calc_probs of the tree's library under two environments, alternating:
#   bash profiles/r05/ab_env.sh "TRX_PROBE_ROWS=1" "TRX_PROBE_ROWS=0"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for k in 1 2; do
for E in "$@"; do
echo "== $E"; env $E python profiles/r05/batch_step.py 4 2>/dev/null | grep "step [234]" | cut -c1-18 | tr '\n' ' '; echo
env $E python profiles/r05/e2e_step.py 3 2>/dev/null | grep "run [123]" | tr '\n' ' '; echo
done
done
