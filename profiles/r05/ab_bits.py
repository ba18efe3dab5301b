"""Bit-level fingerprint of the likelihood kernels' results over a set of shapes, for A/B builds that must not change a
bit (TRX_LIB=<variant> python profiles/r05/ab_bits.py > a.txt; diff a.txt b.txt).  Also whole calc_probs_many runs of
synthetic targets through the launch chains (device mode, seeded): the records' bytes."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np
import torch

import triceratops_amd
from triceratops_amd import _lib, synth

_lib.require_gpu()
rng0 = np.random.default_rng(20251003)


def digest(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]


shapes = [("uniform", 50), ("uniform", 77), ("uniform", 100), ("uniform", 200), ("uniform", 300), ("uniform", 478),
          ("uniform", 2000), ("irregular", 100), ("irregular", 640), ("irregular", 2000), ("wide", 200), ("wide", 1000)]
rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
for kind, nt in shapes:
    rng = np.random.default_rng(1000 + nt)
    if kind == "uniform":
        t = synth.time_grid(nt)
    elif kind == "wide":
        t = np.linspace(-1.5, 1.5, nt)
    else:
        t = np.sort(rng.uniform(-0.25, 0.25, nt))
    ref = synth.reference_tp_row()
    t_d = _lib.dev(t)
    curve = _lib.flux_grid(_lib.MODEL_TP, 0, t_d, _lib.dev(ref), synth.EXPTIME, synth.NSAMPLES, want_secdepth=False)[0].cpu().numpy()[0]
    flux = synth.noisy_light_curve(rng, curve)
    f_d = _lib.dev(flux)
    for fam in synth.FAMILIES[::3]:
        rows = synth.family_rows(rng, fam, rows_n)
        r_d = _lib.dev(rows)
        h = _lib.lnl_batch(fam[1], 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, synth.NSAMPLES).cpu().numpy()
        g = _lib.flux_grid(fam[1], 0, t_d, _lib.dev(rows[:, :2000]), synth.EXPTIME, synth.NSAMPLES, want_secdepth=False)[0].cpu().numpy()
        if len(sys.argv) > 2:
            np.savez(os.path.join(sys.argv[2], "%s_%d_%s.npz" % (kind, nt, fam[0])), h=h, g=g)
        print("%-9s %5d %-6s lnl %s grid %s  (finite %d, min %.6f)" % (kind, nt, fam[0], digest(h), digest(g),
                                                                    int(np.isfinite(h).sum()), float(np.nanmin(h))))
# whole calc_probs_many runs (bounded evaluation, launch chains)
triceratops_amd.set_sampling("device")
for nt, N in ((100, 200_000), (200, 300_000), (60, 100_000)):
    torch.manual_seed(77)
    GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
    jobs = synth.toi_jobs(3, n_time=nt, N=N, seed=5, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                          contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
    triceratops_amd.calc_probs_many(jobs)
    for tg, _ in jobs:
        p = tg.probs
        print("calc_probs %4d points N %7d: lnZ %s  table %s  FPP %.12g" % (
            nt, N, digest(np.asarray(tg.lnZ)), digest(p.select_dtypes("number").to_numpy()), tg.FPP))
