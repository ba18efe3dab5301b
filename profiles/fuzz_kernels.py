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
def raw_stress_rows(rng, n):
    """pytransit-shaped rows far outside the bench's ranges: deep and grazing geometries, k up to 1.5,
    e up to 0.95, periods from 0.3 to 100 d, orbits down to 1.5 stellar radii"""
    k = np.where(rng.random(n) < 0.7, rng.uniform(0.01, 0.3, n), rng.uniform(0.3, 1.5, n))
    a = 10 ** rng.uniform(np.log10(1.5), np.log10(60), n)
    e = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 0.95, n))
    w = rng.uniform(0, 2 * np.pi, n)
    b = rng.uniform(0, 1 + k)
    inc = np.arccos(np.clip(b / (a * (1 - e * e) / (1 + e * np.sin(w))), 0, 1))
    per = 10 ** rng.uniform(np.log10(0.3), 2, n)
    rows = np.stack([k, rng.uniform(-0.02, 0.02, n), per, a, inc, e, w, rng.uniform(0.1, 0.6, n),
                     rng.uniform(0.05, 0.4, n)])
    rows = np.ascontiguousarray(rows[:, a * (1 - e) > 1 + k])
    return rows if rows.shape[1] else raw_stress_rows(rng, n)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
L = _lib.lib()
t_start = time.time()
n_cfg, worst_flux, worst_h, worst_big = 0, 0.0, 0.0, 0.0
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
    model = int(rng.choice([0, 1, 2, _lib.MODEL_RAW]))
    nrow = int(rng.choice([1, 7, 64, 130]))
    if model == 0:
        rows = synth.tp_rows(rng, nrow, bool(rng.integers(2)))
    elif model == _lib.MODEL_RAW:
        rows = raw_stress_rows(rng, 2 * nrow + 8)[:, :nrow]
        nrow = rows.shape[1]
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
        h = (_lib.lnl_batch(model, fl, t_d, f_d, synth.SIGMA, r_d, exptime, S).cpu().numpy()
             if model != _lib.MODEL_RAW else np.zeros(rows.shape[1]))
    finally:
        L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW); L.trx_set_rows_per_wave(0); L.trx_set_stencil(1)
    if model == _lib.MODEL_RAW:
        gw = O.evaluate_pv(t, rows[:7].T, rows[7:].T, exptime, S)
        hw = h
    else:
        gw = O.flux_grid(model, t, rows, companion_is_host=is_host, exptime=exptime, nsamples=S)[0]
        hw = O.lnl_batch(model, t, flux, synth.SIGMA, rows, companion_is_host=is_host, exptime=exptime, nsamples=S)
    ok = np.array_equal(np.isnan(g), np.isnan(gw)) and np.array_equal(np.isinf(h), np.isinf(hw)) and np.array_equal(np.isnan(h), np.isnan(hw))
    # radius ratios above 1 (secondary-eclipse regime of the raw rows): the Mandel-Agol coefficients grow
    # like k^4 and both implementations round differently -- 1e-11 there, 5e-13 for k <= 1 (tests/test_oracle.py)
    big = (rows[0] > 1.0) if model == _lib.MODEL_RAW else np.zeros(rows.shape[1], bool)
    d = np.abs(g - gw)
    df = float(np.nanmax(d[~big])) if (~big).any() and not np.all(np.isnan(d[~big])) else 0.0
    df_big = float(np.nanmax(d[big])) if big.any() and not np.all(np.isnan(d[big])) else 0.0
    worst_big = max(worst_big, df_big)
    fin = np.isfinite(hw)
    dh = float(np.max(np.abs(h[fin] - hw[fin]) / np.maximum(np.abs(hw[fin]), 1.0))) if fin.any() else 0.0
    worst_flux, worst_h = max(worst_flux, df), max(worst_h, dh)
    n_cfg += 1
    # one ulp of a time stamp moves the planet by ulp(t) x a x 2 pi / P stellar radii: on unfolded curves of
    # short-period, wide (a/R ~ 50) raw rows that alone is ~1e-12 in flux, for either implementation
    cond = 0.0
    if model == _lib.MODEL_RAW:
        cond = 4.0 * 2.2e-16 * float(np.max(np.abs(t))) * float(np.max(rows[3] * 2 * np.pi / rows[2]))
    if not ok or df > 5e-13 + cond or df_big > 1e-11 + 10 * cond or dh > 1e-9:
        fails.append((n_time, exptime, S, kind, model, nrow, is_host, below, B, st, df, dh, ok))
        print("FAIL", fails[-1])
        # where, and what the same launch gives with the Kepler stepping off (full solves everywhere) and with every
        # sub-exposure evaluated: a difference that survives both is not the reduced node sets' nor the stepping's
        i, j = np.unravel_index(np.nanargmax(np.where(np.isnan(d), -1, d)), d.shape)
        print("   worst cell: row %d time %d  flux oracle %.15f  kernel %.15f  row parameters %s" % (i, j, gw[i, j], g[i, j], rows[:, i]))
        for what, call in (("Kepler stepping off", lambda v: L.trx_set_kepler_stepping(v)), ("all sub-exposures", lambda v: L.trx_set_supersample_tiers(v))):
            call(0)
            try:
                L.trx_set_cell_packing_below(below); L.trx_set_rows_per_wave(B); L.trx_set_stencil(st)
                g2 = _lib.flux_grid(model, fl, t_d, r_d, exptime, S, want_secdepth=False)[0].cpu().numpy()
            finally:
                call(1)
                L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW); L.trx_set_rows_per_wave(0); L.trx_set_stencil(1)
            print("   %s: worst |dflux| %.3e (that cell: %.3e)" % (what, float(np.nanmax(np.abs(g2 - gw))), abs(g2[i, j] - gw[i, j])))
        if model == _lib.MODEL_RAW and df > 5e-13:
            dd = np.where(big[:, None], 0.0, np.nan_to_num(d))
            r, j = np.unravel_index(np.argmax(dd), dd.shape)
            print("   worst cell: row", repr(rows[:, r].tolist()), "t", repr(float(t[j])), "gpu", repr(float(g[r, j])), "oracle", repr(float(gw[r, j])))
print("%d configurations in %.0f s: worst |dflux| %.2e (k <= 1) / %.2e (k > 1), worst relative |d chi2/2| %.2e, %d failures" % (
    n_cfg, time.time() - t_start, worst_flux, worst_big, worst_h, len(fails)))
sys.exit(1 if fails else 0)
