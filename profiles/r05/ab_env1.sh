#!/bin/bash
# 64-TOI steps (best of 4 steady) and 75-scenario calc_probs under several environments, twice round:  bash profiles/r05/ab_env1.sh "A=1" "B=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for k in 1 2; do
for E in "$@"; do
printf "%-40s" "$E"; env $E python profiles/r05/batch_step.py 4 2>/dev/null | grep "step [1234]" | cut -c9-14 | sort | head -2 | tr '\n' ' '
env $E python profiles/r05/e2e_step.py 3 2>/dev/null | grep "run [123]" | cut -c8-12 | sort | head -2 | tr '\n' ' '; echo
done
done
