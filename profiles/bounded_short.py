"""Bounded evaluation on SHORT light curves (the batched variant of cells_kernel), one lnZ_* call at a time:
GPU time of the call (events on its stream), rows evaluated, rows abandoned, with trx_set_bounded_evaluation 0 / 2.
    python profiles/bounded_short.py"""
import ctypes, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import anchors, triceratops_amd
from triceratops_amd import _lib, fused, synth
from triceratops_amd import marginal_likelihoods as ml
triceratops_amd.set_sampling("device")
L = _lib.lib()
fused.TABLE_ROWS = 1
GOLD = os.path.join(ROOT, "tests", "golden")
N = 1_000_000


def cases():
    for case in ("toi465_nocc", "toi411"):
        stars, t, f, sigma, P = anchors.inputs(case)
        M_s, R_s, Teff, plx = (float(stars[c][0]) for c in ("mass", "rad", "Teff", "plx"))
        yield case, (t, f, sigma, P, M_s, R_s, Teff), plx
    jobs = synth.toi_jobs(2, n_time=200, N=N, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                          contrast_curve_file=None)
    for j, (tg, kw) in enumerate(jobs):
        st = tg.stars
        yield "synth%d" % j, (kw["time"], kw["flux_0"] / 1.0, kw["flux_err_0"], kw["P_orb"], float(st["mass"][0]),
                              float(st["rad"][0]), float(st["Teff"][0])), float(st["plx"][0])


cnt = ctypes.c_ulonglong(0)
for name, base, plx in cases():
    calls = {"TTP": lambda: ml.lnZ_TTP(*base, 0.0, N, True), "TEB": lambda: ml.lnZ_TEB(*base, 0.0, N, True),
             "PTP": lambda: ml.lnZ_PTP(*base, 0.0, plx, None, "TESS", N, True),
             "STP": lambda: ml.lnZ_STP(*base, 0.0, plx, None, "TESS", N, True)}
    for cname, call in calls.items():
        line = "%-12s %-4s n_time %3d:" % (name, cname, len(base[0]))
        for mode in (0, 2):
            L.trx_set_bounded_evaluation(mode)
            best, lnz = 1e9, None
            for rep in range(4):
                torch.manual_seed(11)
                _lib.reset_stats()
                L.trx_pruned_rows(ctypes.byref(cnt), 1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                res = call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
                L.trx_pruned_rows(ctypes.byref(cnt), 1)
            r = res[0] if isinstance(res, tuple) else res
            line += "   bounded %d: %.3f ms, %d rows, %d abandoned, lnZ %.6f" % (mode, best, _lib.STATS["rows"], cnt.value, r["lnZ"])
        print(line, flush=True)
L.trx_set_bounded_evaluation(2)
