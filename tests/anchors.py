"""Runs behind the notebook anchors (tests/test_gpu_notebook_anchors.py, profiles/notebook_anchors.py):
calc_probs on the inputs of the reference's example notebooks, many seeds, one sampling mode.
Fixtures: tests/golden/notebook_anchors.npz (make_anchors.py), toi465_calc_probs.npz, toi465_cc.csv."""
import os

import numpy as np
import pandas as pd

from helpers import GOLD, gold

A = gold("notebook_anchors.npz")
G465 = gold("toi465_calc_probs.npz")
CC465 = os.path.join(GOLD, "toi465_cc.csv")
TRILEGAL = os.path.join(GOLD, "trilegal_synth.csv")
STAR_COLS = ("ID", "Tmag", "Jmag", "Hmag", "Kmag", "ra", "dec", "mass", "rad", "Teff", "plx",
             "fluxratio", "tdepth")
SCENARIOS = [str(s) for s in A["scenarios"]]
# the scenarios whose evidence depends on no TRILEGAL population (synthetic here, a web query in the
# notebooks): target star, bound companion as diluter, bound companion as host
TRILEGAL_FREE = ("TP", "PTP", "STP")

CASES = {
    # name: (ID, mission, stars, time, flux, sigma, P_orb, contrast curve, anchor key)
    "toi465_nocc": dict(ID=270380593, mission="TESS", key="toi465", cc=None),
    "toi465_cc": dict(ID=270380593, mission="TESS", key="toi465", cc=CC465),
    "toi411": dict(ID=100990000, mission="TESS", key="toi411", cc=None),
    "kep10": dict(ID=377780790, mission="Kepler", key="kep10", cc=None),
}


def _stars(src, prefix):
    st = pd.DataFrame({c: src[prefix + c] for c in STAR_COLS})
    st["ID"] = st["ID"].astype(np.int64)
    return st


def inputs(case):
    c = CASES[case]
    if c["key"] == "toi465":
        return (_stars(G465, "real_stars_"), G465["time"], G465["flux"], float(G465["sigma"][0]),
                float(G465["P_orb"][0]))
    k = c["key"]
    return (_stars(A, k + "_stars_"), A[k + "_time"], A[k + "_flux"], float(A[k + "_sigma"][0]),
            float(A[k + "_P_orb"][0]))


def run(case, seed, N=1_000_000, sampling="device"):
    """one calc_probs; returns (lnZ[15], prob[15], FPP, R_p of the TP row)"""
    import torch
    import triceratops_amd
    from triceratops_amd.triceratops import target
    c = CASES[case]
    stars, t, f, sigma, P = inputs(case)
    tg = target(c["ID"], np.array([1]), mission=c["mission"], stars=stars, trilegal_fname=TRILEGAL)
    prev = triceratops_amd.get_sampling()
    triceratops_amd.set_sampling(sampling)
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        tg.calc_probs(t, f, sigma, P, contrast_curve_file=c["cc"], N=N, parallel=True, verbose=0)
    finally:
        triceratops_amd.set_sampling(prev)
    return (np.array(tg.lnZ), tg.probs["prob"].values.copy(), float(tg.FPP),
            float(tg.probs["R_p"].values[0]))


_run_cache = {}


def run_many(case, seeds, N=1_000_000, sampling="device"):
    """(lnZ [runs][15], prob [runs][15], FPP [runs], R_p of the TP row [runs]); kept per (case, seeds, N, mode) so
    that the test files that look at the same runs share them"""
    key = (case, tuple(seeds), N, sampling)
    if key not in _run_cache:
        out = [run(case, s, N, sampling) for s in seeds]
        _run_cache[key] = (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
                           np.array([o[2] for o in out]), np.array([o[3] for o in out]))
    return _run_cache[key]


def notebook(case):
    """(prob[15] of the notebook's single run, its FPP, its TP R_p)"""
    k = CASES[case]["key"]
    return A[k + "_prob"], float(A[k + "_FPP"][0]), float(A[k + "_Rp_TP"][0])


def free_shares(prob):
    """TP : PTP : STP renormalised among themselves (rows of `prob`: runs)"""
    idx = [SCENARIOS.index(s) for s in TRILEGAL_FREE]
    p = np.atleast_2d(prob)[:, idx]
    return p / p.sum(axis=1, keepdims=True)
