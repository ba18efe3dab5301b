#!/bin/bash
# kernel + HIP API time breakdown of the end-to-end run: profiles/stats_e2e.sh <tag> <blend|real>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WHICH=${2:-blend}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG} -- python3 $R/profiles/e2e_toi465.py device $WHICH 3 > $R/gpurun_out/stats_${TAG}.log 2>&1
python3 - <<PY
import csv,glob
for kind in ("kernel_stats","hip_api_stats"):
    fs=glob.glob("$R/gpurun_out/stats_${TAG}/*/*%s.csv"%kind)
    if not fs: print("no",kind); continue
    rows=list(csv.DictReader(open(fs[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("== ${TAG} %s: total %.1f ms, %d calls"%(kind,tot/1e6,sum(int(r["Calls"]) for r in rows)))
    for r in rows[:14]:
        print("   %-60s calls %6s total %9.2f ms avg %8.1f us %5.1f %%"%(r["Name"].replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
tail -3 $R/gpurun_out/stats_${TAG}.log
