#!/bin/bash
# kernel time breakdown of one batch-mode step: profiles/stats_batch.sh <tag> [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG} -- python3 $R/bench.py --mode batch --steps 1 --warmup 1 --no-cpu-baseline --pmc off "$@" > $R/gpurun_out/stats_${TAG}.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/stats_${TAG}/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== ${TAG}: all kernels %.1f ms, %d launches"%(tot/1e6,sum(int(r["Calls"]) for r in rows)))
for r in rows[:12]:
    print("   %-56s calls %6s total %9.2f ms avg %8.1f us %5.1f %%"%(r["Name"].replace("(anonymous namespace)::","")[:56],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
tail -c 600 $R/gpurun_out/stats_${TAG}.log | grep -o '"ms_per_step": [0-9.]*'
