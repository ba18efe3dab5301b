#!/bin/bash
# mean duration of the chain's kernels on ONE stream (16 TOIs, two steps) under several environments:
#   bash profiles/r05/trace_env.sh "A=1" "B=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
k=0
for E in "$@"; do
  k=$((k+1))
  rm -rf /tmp/trace_env_$k
  export $E
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_env_$k -o t -- python3 $R/profiles/r05/batch_step.py 2 16 1000000 1 > /tmp/trace_env_$k.log 2>&1
  for v in $E; do unset ${v%%=*}; done
  python3 - "$E" /tmp/trace_env_$k <<'PY'
import csv, glob, sys, collections
E, d = sys.argv[1:3]
f = glob.glob(d + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                      # (skip the first step)
tot, cnt = collections.Counter(), collections.Counter()
part = 0
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "_star" not in n:
        continue
    n = n.replace("void ", "")
    if "draw_kernel_star" in n:
        part = 0
    if "cells_kernel_star" in n:
        part += 1
        n += " #%d" % part
    tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[n] += 1
chain = sum(tot.values()) / max(cnt[[k for k in cnt if "draw" in k][0]], 1)
print("%-44s chain %7.1f us | %s" % (E, chain, "  ".join("%s %.0f" % (k.replace("_kernel_star", "").replace("<false, false>", ""), tot[k] / cnt[k]) for k in tot)))
PY
done
