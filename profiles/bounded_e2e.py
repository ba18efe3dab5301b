"""calc_probs wall-clock with and without the bounded evaluation on the notebook targets (device sampling, N = 1e6)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import anchors, triceratops_amd
from triceratops_amd import _lib
from triceratops_amd.triceratops import target
triceratops_amd.set_sampling("device")
L = _lib.lib()
for case in ("toi465_cc", "toi411", "kep10"):
    c = anchors.CASES[case]; stars, t, f, sigma, P = anchors.inputs(case)
    for mode in (1, 0, 1, 0):
        L.trx_set_bounded_evaluation(mode)
        best = 9
        for rep in range(4):
            tg = target(c["ID"], np.array([1]), mission=c["mission"], stars=stars, trilegal_fname=anchors.TRILEGAL)
            np.random.seed(3); torch.manual_seed(3)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tg.calc_probs(t, f, sigma, P, contrast_curve_file=c["cc"], N=1_000_000, parallel=True, verbose=0)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%-12s n_time %4d bounded %d: %.4f s  FPP %.6g" % (case, t.size, mode, best, tg.FPP))
