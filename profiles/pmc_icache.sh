#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for NT in 100 2000; do
rm -rf /tmp/pmc_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pmc_ic -- python3 $R/profiles/cells_once.py $NT 100000 $( [ $NT = 100 ] && echo cells || echo rows ) > /tmp/pmc_ic.log 2>&1
python3 - $NT <<'PY'
import csv,glob,sys,collections
tot=collections.Counter()
for f in glob.glob("/tmp/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cells_kernel<0" in r["Kernel_Name"].replace(" ",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[1], dict(tot))
PY
done
