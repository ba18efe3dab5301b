"""Randomised stress of the likelihood kernels against the CPU oracle (not part of the test suite: the
oracle is test infrastructure, this script is a one-off checker like the tests).
Random light-curve lengths, time grids (uniform dense / uniform sparse / jittered / irregular /
unfolded), exposures, sub-exposure counts, models, rows-per-wave, variant thresholds, stencil on/off.
python profiles/fuzz_kernels.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from triceratops_amd import _lib, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
L = _lib.lib()
t_start = time.time()
n_cfg, worst_flux, worst_h = 0, 0.0, 0.0
fails = []
while time.time() - t_start < budget:
    n_time = int(rng.choice([1, 2, 63, 64, 65, 100, 200, 319, 320, 321, 500, 777, 1024, 1025, 1500, 2000, 2500, 3000]))
    exptime = float(rng.choice([0.00139, 0.0204, 0.0, 0.005]))
    S = int(rng.choice([1, 2, 7, 8, 12, 20, 21, 50])) if exptime > 0 else 1
    kind = rng.choice(["dense", "sparse", "jitter", "irregular", "unfolded"])
    if kind == "dense" and exptime > 0:
        dt = exptime * float(rng.choice([0.05, 0.12, 0.15, 0.18, 0.25, 0.3, 0.33]))
        t = np.linspace(-0.5 * dt * (n_time - 1), 0.5 * dt * (n_time - 1), n_time) + float(rng.uniform(-0.01, 0.01))
    elif kind == "sparse":
        t = np.linspace(-0.3, 0.3, n_time) if n_time > 1 else np.array([0.01])
    elif kind == "jitter":
        t = np.sort(np.linspace(-0.2, 0.2, n_time) + rng.uniform(-1e-4, 1e-4, n_time))
    elif kind == "irregular":
        t = np.sort(rng.uniform(-0.4, 0.4, n_time))
    else:
        t = np.sort(rng.uniform(-20.0, 20.0, n_time))
    model = int(rng.choice([0, 1, 2]))
    nrow = int(rng.choice([1, 7, 64, 130]))
    if model == 0:
        rows = synth.tp_rows(rng, nrow, bool(rng.integers(2)))
    else:
        rows = synth.eb_rows(rng, nrow, model == 2, bool(rng.integers(2)))
    is_host = bool(rng.integers(2))
    flux = 1.0 + rng.normal(0, synth.SIGMA, n_time)
    below = int(rng.choice([0, 1 << 30, _lib.CELL_PACKING_BELOW]))
    B = int(rng.choice([0, 1, 2, 5, 22]))
    st = int(rng.integers(2))
    L.trx_set_cell_packing_below(below); L.trx_set_rows_per_wave(B); L.trx_set_stencil(st)
    try:
        t_d, f_d, r_d = _lib.dev(t), _lib.dev(flux), _lib.dev(rows)
        fl = _lib.FLAG_COMPANION_IS_HOST if is_host else 0
        g = _lib.flux_grid(model, fl, t_d, r_d, exptime, S, want_secdepth=False)[0].cpu().numpy()
        h = _lib.lnl_batch(model, fl, t_d, f_d, synth.SIGMA, r_d, exptime, S).cpu().numpy()
    finally:
        L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW); L.trx_set_rows_per_wave(0); L.trx_set_stencil(1)
    gw = O.flux_grid(model, t, rows, companion_is_host=is_host, exptime=exptime, nsamples=S)[0]
    hw = O.lnl_batch(model, t, flux, synth.SIGMA, rows, companion_is_host=is_host, exptime=exptime, nsamples=S)
    ok = np.array_equal(np.isnan(g), np.isnan(gw)) and np.array_equal(np.isinf(h), np.isinf(hw)) and np.array_equal(np.isnan(h), np.isnan(hw))
    df = float(np.nanmax(np.abs(g - gw))) if g.size and not np.all(np.isnan(g)) else 0.0
    fin = np.isfinite(hw)
    dh = float(np.max(np.abs(h[fin] - hw[fin]) / np.maximum(np.abs(hw[fin]), 1.0))) if fin.any() else 0.0
    worst_flux, worst_h = max(worst_flux, df), max(worst_h, dh)
    n_cfg += 1
    if not ok or df > 5e-13 or dh > 1e-9:
        fails.append((n_time, exptime, S, kind, model, nrow, is_host, below, B, st, df, dh, ok))
        print("FAIL", fails[-1])
print("%d configurations in %.0f s: worst |dflux| %.2e, worst relative |d chi2/2| %.2e, %d failures" % (
    n_cfg, time.time() - t_start, worst_flux, worst_h, len(fails)))
sys.exit(1 if fails else 0)
