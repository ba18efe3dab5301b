"""model evaluations planned per cell, by n_time (census knob): python profiles/node_census2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
L = _lib.lib()
for n_time in (100, 200, 2000):
    rng = np.random.default_rng(synth.SEED)
    t_d = _lib.dev(synth.time_grid(n_time))
    tot = {}
    ncell = 0
    L.trx_set_debug_node_counts(1)
    try:
        for fam in synth.FAMILIES:
            rows = _lib.dev(synth.family_rows(rng, fam, 2000))
            c, _ = _lib.flux_grid(fam[1], 0, t_d, rows, synth.EXPTIME, 20, False)
            v, k = torch.unique(c, return_counts=True)
            for a, b in zip(v.cpu().numpy(), k.cpu().numpy()):
                tot[int(a)] = tot.get(int(a), 0) + int(b)
            ncell += c.numel()
    finally:
        L.trx_set_debug_node_counts(0)
    ev = sum(a * b for a, b in tot.items()) / ncell
    print("n_time %4d: evaluations per cell %.2f;" % (n_time, ev), " ".join("%d nodes %.1f%% (%.0f%% of evals)" % (a, 100 * b / ncell, 100 * a * b / ncell / ev if ev else 0) for a, b in sorted(tot.items())))
