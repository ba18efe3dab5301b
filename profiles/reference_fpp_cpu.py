"""The REFERENCE's own calc_probs (imported from /root/reference under the shims of
tests/golden/make_golden.py: oracle QuadraticModel at the pytransit seam) at N = 1e6 on the
notebook inputs, on the CPU of the build container.  Tells apart "our device path differs from the
current reference code" from "the notebooks were made by an older release / by pytransit itself".
    python profiles/reference_fpp_cpu.py toi465_nocc 5 [N] [first seed] > profiles/r03/reference_fpp_cpu.txt"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

mg.install_shims()
rtr = mg.import_reference_target()
import anchors  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "toi465_nocc"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000
first = int(sys.argv[4]) if len(sys.argv) > 4 else 1000          # first seed
c = anchors.CASES[case]
stars, t, f, sigma, P = anchors.inputs(case)
for seed in range(first, first + runs):
    tg = object.__new__(rtr.target)
    tg.ID, tg.mission, tg.sectors = c["ID"], c["mission"], np.array([1])
    tg.search_radius, tg.N_pix, tg.trilegal_fname, tg.trilegal_url = 10, 22, anchors.TRILEGAL, None
    tg.stars = stars
    np.random.seed(seed)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        tg.calc_probs(t, f, sigma, P, contrast_curve_file=c["cc"], N=N, parallel=True, verbose=0)
    pr = tg.probs["prob"].values
    print("%s seed %d N %d: FPP %.5f  TP %.4f PTP %.4f STP %.5f DTP %.5f  lnZ TP %.3f PTP %.3f STP %.3f  Rp %.3f  (%.0f s)"
          % (case, seed, N, tg.FPP, pr[0], pr[3], pr[6], pr[9], tg.lnZ[0], tg.lnZ[3], tg.lnZ[6],
             tg.probs["R_p"].values[0], time.perf_counter() - t0), flush=True)
