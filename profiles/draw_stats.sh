#!/bin/bash
# draw_kernel durations by position in a TOI's 12 calls (one stream, nothing overlapping): profiles/draw_stats.sh <tag> [lib]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
if [ -n "$2" ]; then export TRX_LIB=$R/profiles/ab_libs/libtrx_$2.so; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/draw_${TAG} -- python3 $R/profiles/draw_times.py > $R/gpurun_out/draw_${TAG}.log 2>&1
python3 - <<PY
import csv,glob,statistics,collections
f=glob.glob("$R/gpurun_out/draw_${TAG}/*/*kernel_trace.csv")[0]
allrows=list(csv.DictReader(open(f)))
names=["TP","EB","PTP","PEB","STP","SEB","DTP","DEB","BTP","BEB","NTP","NEB"]
for kern in ("draw_kernel","compact_fill_kernel"):
    rows=[r for r in allrows if kern in r["Kernel_Name"]]
    if not rows: continue
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    rows=rows[-48:]
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
    per=collections.defaultdict(list)
    for j,x in enumerate(d): per[j%12].append(x)
    print("== ${TAG}: %s us per call, N = 1e6: "%kern+"  ".join("%s %.0f"%(names[j],statistics.mean(per[j])) for j in range(12))+"   | mean %.1f"%statistics.mean(d))
PY
