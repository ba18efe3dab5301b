#!/bin/bash
# the GPU suite with eight busy-loop processes beside it (a slow / contended host must not
# fail it: no test asserts an absolute wall-clock any more).   bash profiles/r05/suite_on_a_busy_box.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
pids=""
for i in 1 2 3 4 5 6 7 8; do
  ( while :; do :; done ) &
  pids="$pids $!"
done
echo "busy loops: $pids; nproc $(nproc)"
start=$(date +%s)
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|host path|rows abandoned|error" | tail -20
echo "suite wall-clock: $(( $(date +%s) - start )) s"
kill $pids
