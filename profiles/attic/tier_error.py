"""Reduced-node exposure averaging vs all S sub-exposures, same kernel, same inputs:
max |flux difference| per scenario family and the relative change of chi^2/2.
usage: python profiles/tier_error.py [rows_per_family]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(7)
L = _lib.lib()
worst = 0.0
for n_time in (2000, 200):
    t = synth.time_grid(n_time); t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    for fam in synth.FAMILIES:
        rows = _lib.dev(synth.family_rows(rng, fam, n))
        flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
        out = {}
        for on in (1, 0):
            L.trx_set_supersample_tiers(on)
            g, _ = _lib.flux_grid(fam[1], flags, t_d, rows, synth.EXPTIME, 20, False)
            h = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
            out[on] = (g, h)
        L.trx_set_supersample_tiers(1)
        d = (out[1][0] - out[0][0]).abs()
        fin = torch.isfinite(out[0][1])
        rel = ((out[1][1][fin] - out[0][1][fin]).abs() / out[0][1][fin].abs()).max().item() if fin.any() else 0.0
        same_inf = bool(torch.equal(torch.isfinite(out[1][1]), fin))
        frac = (d > 0).double().mean().item()
        print("n_time %4d %-8s max|dflux| %.2e  p99.9 %.2e  cells changed %.3f  max rel d(chi2/2) %.2e  masks equal %s"
              % (n_time, fam[0], d.max().item(), torch.quantile(d.flatten()[::7], 0.999).item(), frac, rel, same_inf))
        worst = max(worst, d.max().item())
print("worst", worst)

# ---- stress: raw pytransit-shaped rows far outside the bench's parameter ranges ----------------
print("stress (MODEL_RAW)")
n = 4096
for exptime, S, span in ((0.00139, 20, 0.3), (0.0204, 20, 0.6), (0.0204, 50, 0.6), (0.00139, 12, 0.3), (0.00139, 20, None)):
    k = np.where(rng.random(n) < 0.7, rng.uniform(0.01, 0.3, n), rng.uniform(0.3, 1.5, n))
    a = 10 ** rng.uniform(np.log10(1.5), np.log10(60), n)
    e = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 0.95, n))
    w = rng.uniform(0, 2 * np.pi, n)
    b = rng.uniform(0, 1 + k)
    r_tr = (1 - e * e) / (1 + e * np.sin(w))                 # r/a at inferior conjunction
    inc = np.arccos(np.clip(b / (a * r_tr), 0, 1))
    per = 10 ** rng.uniform(np.log10(0.3), 2, n)
    ok = a * (1 - e) > 1 + k                                   # no contact orbits
    t0 = rng.uniform(-0.02, 0.02, n)
    u1, u2 = rng.uniform(0.1, 0.6, n), rng.uniform(0.05, 0.4, n)
    rows = np.stack([k, t0, per, a, inc, e, w, u1, u2])[:, ok]
    if span is None:                                           # unfolded: several periods of the shortest orbit
        t = np.sort(rng.uniform(-3.0, 3.0, 3000))
    else:
        t = np.linspace(-span, span, 1500)
    t_d, rows_d = _lib.dev(t), _lib.dev(rows)
    g = {}
    for on in (1, 0):
        L.trx_set_supersample_tiers(on)
        g[on], _ = _lib.flux_grid(_lib.MODEL_RAW, 0, t_d, rows_d, exptime, S, False)
    L.trx_set_supersample_tiers(1)
    d = (g[1] - g[0]).abs()
    d = torch.where(torch.isnan(g[0]) & torch.isnan(g[1]), torch.zeros_like(d), d)
    i = int(torch.argmax(d.max(dim=1).values))
    print("exptime %.5f S %2d span %s rows %d  max|dflux| %.2e  changed %.3f  in transit %.3f  worst row k=%.3f a=%.2f e=%.2f P=%.2f b=%.2f"
          % (exptime, S, span, rows.shape[1], d.max().item(), (d > 0).double().mean().item(),
             (g[0] < 1).double().mean().item(), rows[0, i], rows[3, i], rows[5, i], rows[2, i],
             rows[3, i] * np.cos(rows[4, i]) * (1 - rows[5, i] ** 2) / (1 + rows[5, i] * np.sin(rows[6, i]))))
    worst = max(worst, d.max().item())
print("worst incl. stress", worst)
