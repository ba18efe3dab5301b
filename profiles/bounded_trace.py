"""kernel trace of one lnZ_TTP call (TOI-465.01, N = 1e6) with the bounded evaluation off / on for batches:
    rocprofv3 --kernel-trace -- python3 profiles/bounded_trace.py ; then profiles/bounded_trace.py --summary <csv>"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
if "--summary" in sys.argv:
    import csv
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--summary") + 1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last two calls: mode 0 then mode 2 (markers: draw_kernel starts a call)
    starts = [i for i, r in enumerate(rows) if "draw_kernel" in r["Kernel_Name"]]
    for s0, s1, tag in ((starts[-2], starts[-1], "bounded 0"), (starts[-1], len(rows), "bounded 2")):
        t0 = int(rows[s0]["Start_Timestamp"])
        print("== %s" % tag)
        for r in rows[s0:s1]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
            print("   %-58s start %8.1f us  dur %7.1f us  grid %s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3,
                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
    sys.exit(0)
import numpy as np, torch
import anchors, triceratops_amd
from triceratops_amd import _lib, fused
from triceratops_amd import marginal_likelihoods as ml
case = sys.argv[1] if len(sys.argv) > 1 else "toi465_nocc"
triceratops_amd.set_sampling("device")
L = _lib.lib()
fused.TABLE_ROWS = 1
stars, t, f, sigma, P = anchors.inputs(case)
M_s, R_s, Teff = (float(stars[c][0]) for c in ("mass", "rad", "Teff"))
for mode in (0, 2, 0, 2, 0, 2):
    L.trx_set_bounded_evaluation(mode)
    torch.manual_seed(11)
    (ml.lnZ_TEB if os.environ.get("TRACE_EB") else ml.lnZ_TTP)(t, f, sigma, P, M_s, R_s, Teff, 0.0, 1_000_000, True)
    torch.cuda.synchronize()
