#!/usr/bin/env python3
"""tests/golden/reference_full.npz: the REFERENCE's own calc_probs at N = 1e6 (and 1e5), seeded, with every
scenario's lnZ, probability and best-fit row at FULL precision (reference_runs.npz keeps only what a log line
printed to 1e-3).  The reference is imported from /root/reference under the shims of make_golden.py (oracle
QuadraticModel at the pytransit seam), in the build container only; ~90 s per run at N = 1e6.

These runs pin the PRODUCTION chain of this repository -- set_sampling("numpy-device") under calc_probs feeds
numpy's seeded uniforms to trx_star_enqueue with the bounded evaluation at its default -- against the reference
value for value at the size the bounded evaluation's probe pass actually runs at
(tests/test_gpu_production_pin.py).

    python tests/golden/make_reference_full.py
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

mg.install_shims()
rtr = mg.import_reference_target()
import anchors  # noqa: E402

RUNS = [("toi465_nocc", 1000, 1_000_000), ("toi411", 1000, 1_000_000), ("toi465_cc", 1001, 1_000_000),
        ("toi465_nocc", 2000, 100_000), ("toi411", 2001, 100_000), ("kep10", 2002, 100_000)]
COLS = ("M_s", "R_s", "P_orb", "inc", "b", "ecc", "w", "R_p", "M_EB", "R_EB", "prob")
out = {"runs": np.array(["%s_%d_%d" % r for r in RUNS])}
for case, seed, N in RUNS:
    c = anchors.CASES[case]
    stars, t, f, sigma, P = anchors.inputs(case)
    tg = object.__new__(rtr.target)
    tg.ID, tg.mission, tg.sectors = c["ID"], c["mission"], np.array([1])
    tg.search_radius, tg.N_pix, tg.trilegal_fname, tg.trilegal_url = 10, 22, anchors.TRILEGAL, None
    tg.stars = stars
    np.random.seed(seed)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        tg.calc_probs(t, f, sigma, P, contrast_curve_file=c["cc"], N=N, parallel=True, verbose=0)
    key = "%s_%d_%d" % (case, seed, N)
    out[key + "_lnZ"] = np.array(tg.lnZ, dtype=np.float64)
    out[key + "_FPP"] = np.array([tg.FPP])
    out[key + "_NFPP"] = np.array([tg.NFPP])
    for col in COLS:
        out[key + "_" + col] = tg.probs[col].values.astype(np.float64)
    for a in ("u1", "u2", "fluxratio_EB", "fluxratio_comp"):
        out[key + "_" + a] = np.array(getattr(tg, a), dtype=np.float64)
    print("%s: FPP %.6g NFPP %.3g lnZ TP %.9f (%.0f s)" % (key, tg.FPP, tg.NFPP, tg.lnZ[0], time.perf_counter() - t0),
          flush=True)
    np.savez_compressed(os.path.join(HERE, "reference_full.npz"), **out)
