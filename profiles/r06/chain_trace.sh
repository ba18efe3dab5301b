#!/bin/bash
# kernel trace of 64-TOI steps on ONE stream (the kernels of a chain back to back): time per kernel and per pass
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T=${1:-chain}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${T}_prof -o batch -- python3 $R/profiles/r05/batch_step.py 2 64 1000000 1 > $O/${T}_prof.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/${T}_prof/**/batch_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the last step: everything after the last gap > 3 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e for _, e, _ in ev[max(0, i - 50):i]) > 3_000_000:
        cut = i
seg = ev[cut:]
tot = collections.defaultdict(lambda: [0, 0.0])
part = 0
for s, e, n in seg:
    name = n.replace("(anonymous namespace)::", "").split("(")[0]
    if "cells_kernel_star" in name:
        part = part % 3 + 1
        name = "cells_kernel_star part %d (%s)" % (part, ["pilot", "probe pass", "survivors"][part - 1])
    tot[name][0] += 1
    tot[name][1] += (e - s) / 1e6
span = (max(e for _, e, _ in seg) - seg[0][0]) / 1e6
print("last step: %d dispatches, span %.1f ms, kernel-time sum %.1f ms" % (len(seg), span, sum(v[1] for v in tot.values())))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s %4d launches %8.2f ms  (%.1f us each)" % (k[:60], v[0], v[1], 1e3 * v[1] / v[0]))
PY
grep step $O/${T}_prof.log
