#!/bin/bash
# grid-mode kernel numbers of several builds of the library, alternating, in one job:
#   bash profiles/r06/ab_kernel.sh <tag> libA.so libB.so ...      ("tree" = triceratops_amd/libtrx.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
T=$1; shift
for rep in 1 2; do
for L in "$@"; do
  P=$R/profiles/ab_libs/$L; [ "$L" = tree ] && P=$R/triceratops_amd/libtrx.so
  TRX_LIB=$P python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-batch-leg --pmc off 2>/dev/null | tail -1 > $O/ab_${T}_${L}_$rep.json
  python - <<PY
import json
d=json.load(open("$O/ab_${T}_${L}_$rep.json"))
r=d["roofline"]; s=d["shapes"]
print("%-28s rep $rep  cfg1 %.3f ms (%.3g/s, evals/cell %.3f)  all-sub %.2f ms  gauss %.3f ms | n100 %.4f ms  n200 %.4f ms  irregular %.3f ms" % (
  "$L", r["mean_launch_ms"], d["value"], r["model_evaluations_per_cell"], r.get("all_subexposures",{}).get("mean_launch_ms",0), r.get("gauss_nodes_only",{}).get("mean_launch_ms",0),
  s["n100"]["mean_launch_ms"], s["n200"]["mean_launch_ms"], s["n2000_irregular"]["mean_launch_ms"]))
PY
done
done | tee $O/ab_${T}.txt
