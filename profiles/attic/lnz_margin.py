"""|lnZ - golden| of every seeded lnZ_* case on the GPU (margin of tests/test_gpu_golden.py's 1e-9)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_golden as T
from triceratops_amd import marginal_likelihoods as ml
worst = 0
for case in T.CASES:
    name, variant = case.split("_")
    P = [2.5, 4.0] if variant == "range" else 3.3
    cc = os.path.join(T.GOLD, "contrast_curve_synth.csv") if variant == "ccJ" else None
    parallel = variant != "serial"
    np.random.seed(int(T.G[case + "_seed"][0]))
    res = T._call(ml, name, P, int(T.G["N"][0]) if parallel else 300, parallel, cc, "J" if cc else "TESS")
    for i, d in enumerate(res if isinstance(res, tuple) else (res,)):
        want = T.G["%s_lnZ%d" % (case, i)][0]
        if np.isfinite(want):
            e = abs(d["lnZ"] - want)
            worst = max(worst, e)
            if e > 1e-10:
                print("%-14s %d lnZ %.6f  |diff| %.3e" % (case, i, want, e))
print("worst", worst)
