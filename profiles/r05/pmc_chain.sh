#!/bin/bash
# PMC counters of the kernels of a launch chain (8 TOIs, one stream, two steps):  bash profiles/r05/pmc_chain.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-chain}
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${T}_$i
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_${T}_$i -- python3 $R/profiles/r05/batch_step.py 1 8 1000000 1 > /tmp/pmc_${T}_$i.log 2>&1
done
python3 - $T <<'PY' > $R/gpurun_out/pmc_${1:-chain}_summary.txt 2>&1
import csv, glob, sys, collections
T = sys.argv[1]
tot = collections.defaultdict(collections.Counter)
calls = collections.Counter()
for i in (1, 2, 3, 4):
    for f in glob.glob("/tmp/pmc_%s_%d/**/*counter_collection.csv" % (T, i), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and r["Counter_Name"] == "SQ_WAVES":
                calls[k] += 1
# the passes of the bounded evaluation one by one (dispatch order: pilot, probe pass, survivors, per chain)
per = collections.defaultdict(collections.Counter)
for f in glob.glob("/tmp/pmc_%s_1/**/*counter_collection.csv" % T, recursive=True):
    for r in csv.DictReader(open(f)):
        if "cells_kernel_star" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)
part = collections.defaultdict(collections.Counter)
for n_, d in enumerate(ids):
    for c, v in per[d].items():
        part[n_ % 3][c] += v
for q in range(3):
    a = part[q]
    m = max(len(ids) // 3, 1)
    print("cells_kernel_star pass %d (%s): per launch waves %.3g VALU %.4g SALU %.4g | lanes %.2f | VALU active / wave cycles %.3f"
          % (q + 1, ("pilot", "probe pass", "survivors")[q], a["SQ_WAVES"] / m, a["SQ_INSTS_VALU"] / m, a["SQ_INSTS_SALU"] / m,
             a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_ACTIVE_INST_VALU"] * 64, 1), a["SQ_ACTIVE_INST_VALU"] / max(a["SQ_WAVE_CYCLES"], 1)))
for k, a in sorted(tot.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    if a["SQ_INSTS_VALU"] < 1e6:
        continue
    n = max(calls[k], 1)
    print("%-42s launches %4d | per launch: waves %.3g VALU %.4g SALU %.4g LDS %.4g VMEM rd/wr %.3g/%.3g | lanes %.2f | VALU active / wave cycles %.3f | wait-any / wave cycles %.3f (LDS wait %.3f) | fetch %.1f MB write %.1f MB"
          % (k, n, a["SQ_WAVES"] / n, a["SQ_INSTS_VALU"] / n, a["SQ_INSTS_SALU"] / n, a["SQ_INSTS_LDS"] / n, a["SQ_INSTS_VMEM_RD"] / n, a["SQ_INSTS_VMEM_WR"] / n,
             a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_ACTIVE_INST_VALU"] * 64, 1), a["SQ_ACTIVE_INST_VALU"] / max(a["SQ_WAVE_CYCLES"], 1),
             a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1), a["SQ_WAIT_INST_LDS"] / max(a["SQ_WAVE_CYCLES"], 1),
             a["FETCH_SIZE"] / n / 1024 * 2 / 1e3 * 1.024, a["WRITE_SIZE"] / n / 1024 / 1e3 * 1.024))
PY
cat $R/gpurun_out/pmc_${T}_summary.txt; tail -3 /tmp/pmc_${T}_1.log
