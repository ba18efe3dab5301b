import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from triceratops_amd import _lib, synth
n_time, model = 100, 1
rng = np.random.default_rng(900 + n_time)
t = np.sort(rng.uniform(-0.25, 0.25, n_time))
n = 1031
rows = synth.eb_rows(rng, n, False, True)
flux = 1.0 + rng.normal(0, synth.SIGMA, n_time)
t_d, f_d, r_d = _lib.dev(t), _lib.dev(flux), _lib.dev(rows)
L = _lib.lib()
wg, ws = O.flux_grid(model, t, rows[:, :64].copy())
wh = O.lnl_batch(model, t, flux, synth.SIGMA, rows[:, :64].copy())
for name, below, tiers in (("rows", 0, 1), ("cells", 1 << 30, 1), ("rows-notiers", 0, 0), ("cells-notiers", 1 << 30, 0)):
    L.trx_set_cell_packing_below(below); L.trx_set_supersample_tiers(tiers)
    g, s = _lib.flux_grid(model, 0, t_d, r_d, synth.EXPTIME, synth.NSAMPLES)
    h = _lib.lnl_batch(model, 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, synth.NSAMPLES).cpu().numpy()[:64]
    g = g.cpu().numpy()[:64]
    d = np.abs(g - wg)
    i, j = np.unravel_index(np.nanargmax(d), d.shape)
    fin = np.isfinite(wh)
    rel = np.abs(h[fin] / wh[fin] - 1)
    print(name, "max |dflux| %.3g at row %d cell %d (t=%.6f, oracle %.12f got %.12f)" % (d[i, j], i, j, t[j], wg[i, j], g[i, j]),
          "max rel dh %.3g at %d" % (rel.max(), np.flatnonzero(fin)[rel.argmax()]))
    r = np.flatnonzero(fin)[rel.argmax()]
    dr = np.abs(g[r] - wg[r]); print("   row", r, "params", rows[:, r], "worst cells", np.argsort(dr)[-3:], dr[np.argsort(dr)[-3:]], "h", h[r], wh[r])
    # chi2 from the grid itself
    hg = 0.5 * np.sum((flux - g[r]) ** 2 / synth.SIGMA ** 2); print("   chi2/2 from this kernel's grid:", hg)
L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW); L.trx_set_supersample_tiers(1)
