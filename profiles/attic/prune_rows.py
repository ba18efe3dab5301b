"""bounded evaluation row by row: trx_lnl_batch with and without it on the bench families"""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from triceratops_amd import _lib, synth
L = _lib.lib()
n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = np.random.default_rng(3)
t = synth.time_grid(n_time); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
for fam in synth.FAMILIES[:6]:
    rows = _lib.dev(synth.family_rows(rng, fam, n))
    flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
    L.trx_set_debug_bounded_lnl(0)
    full = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20).cpu().numpy()
    L.trx_set_debug_bounded_lnl(1)
    cnt = ctypes.c_ulonglong(0); L.trx_pruned_rows(ctypes.byref(cnt), 1)
    got = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20).cpu().numpy()
    L.trx_pruned_rows(ctypes.byref(cnt), 1)
    L.trx_set_debug_bounded_lnl(0)
    fin = np.isfinite(full)
    hmin = full[fin].min()
    exact = np.isclose(got, full, rtol=1e-11, atol=0) | (~fin & ~np.isfinite(got))
    bound_ok = (~exact) & (got <= full * (1 + 1e-9)) & (got > hmin) & (got > hmin + 90 - 1e-6)
    bad = ~(exact | bound_ok)
    print("%-8s n_time %d: %d rows, %d abandoned (counter %d), min %.3f; violations %d" % (fam[0], n_time, n, int((~exact).sum()), cnt.value, hmin, int(bad.sum())))
    for i in np.argwhere(bad).ravel()[:5]:
        print("     row %d: full %.6f bounded %.6f" % (i, full[i], got[i]))
