#!/bin/bash
# lane activity and VALU instruction counts of cells_kernel for library builds: profiles/pmc_activity.sh <n_time> <rows> <cells|rows> lib...
R=${GRAFT_REPO_ROOT:-/root/repo}
NT=$1; NR=$2; WHICH=$3; shift 3
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$R/profiles/ab_libs/libtrx_$L.so; fi
  rm -rf /tmp/pmc_act_$L
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_act_$L -- python3 $R/profiles/cells_once.py $NT $NR $WHICH > /tmp/pmc_act_$L.log 2>&1
  python3 - "$L" <<'PY'
import csv,glob,sys,collections
L=sys.argv[1]
tot=collections.Counter(); n=0
for f in glob.glob("/tmp/pmc_act_%s/**/*counter_collection.csv"%L, recursive=True):
    for r in csv.DictReader(open(f)):
        if "cells_kernel<0" in r["Kernel_Name"].replace(" ",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
a=tot
if a["SQ_ACTIVE_INST_VALU"]:
    print("[%s] VALU insts %.4g  SALU %.4g  lane activity %.3f  VALU busy %.3f" % (L, a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"],
          a["SQ_THREAD_CYCLES_VALU"]/(a["SQ_ACTIVE_INST_VALU"]*64), a["SQ_ACTIVE_INST_VALU"]*4/max(a["SQ_WAVE_CYCLES"],1)*0+a["SQ_ACTIVE_INST_VALU"]/max(a["SQ_BUSY_CYCLES"],1)))
else:
    print("[%s] no counters"%L)
PY
done
