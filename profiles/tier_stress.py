"""Wider adversarial check of the reduced-node exposure average: raw rows over very wide ranges,
several cadences / S, reduced nodes vs all sub-exposures on the same kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
worst = 0.0
for exptime, S, span, npts in ((0.00139, 20, 0.5, 3000), (0.0204, 20, 1.0, 2000), (0.0417, 30, 2.0, 2000),
                               (0.000231, 20, 0.2, 3000), (0.00139, 100, 0.4, 1500), (0.0204, 9, 0.8, 1500)):
    n = 8192
    k = 10 ** rng.uniform(np.log10(0.003), np.log10(1.6), n)
    a = 10 ** rng.uniform(np.log10(1.3), np.log10(300), n)
    e = np.where(rng.random(n) < 0.4, 0.0, rng.uniform(0, 0.97, n))
    w = rng.uniform(0, 2 * np.pi, n)
    b = rng.uniform(0, 1 + k) * np.where(rng.random(n) < 0.2, 1.0, rng.uniform(0.9, 1.0, n))   # many grazing
    inc = np.arccos(np.clip(b / (a * (1 - e * e) / (1 + e * np.sin(w))), 0, 1))
    per = 10 ** rng.uniform(np.log10(0.25), 3, n)
    ok = a * (1 - e) > 1 + k
    rows = np.ascontiguousarray(np.stack([k, rng.uniform(-0.05, 0.05, n), per, a, inc, e, w,
                                          rng.uniform(0.0, 0.8, n), rng.uniform(-0.1, 0.5, n)])[:, ok])
    t = np.sort(rng.uniform(-span, span, npts))
    g = {}
    for on in (1, 0):
        L.trx_set_supersample_tiers(on)
        g[on] = _lib.flux_grid(_lib.MODEL_RAW, 0, _lib.dev(t), _lib.dev(rows), exptime, S, False)[0]
    L.trx_set_supersample_tiers(1)
    d = (g[1] - g[0]).abs()
    d = torch.where(torch.isnan(g[0]) & torch.isnan(g[1]), torch.zeros_like(d), d)
    dep = (1 - g[0]).clamp(min=1e-3)
    i = int(torch.argmax(d.max(dim=1).values))
    print("exptime %.6f S %3d rows %d: max|dflux| %.2e  max rel-to-depth %.2e  changed %.3f  occulted %.3f  worst k=%.4f a=%.2f e=%.2f P=%.2f"
          % (exptime, S, rows.shape[1], d.max().item(), (d / dep).max().item(), (d > 0).double().mean().item(),
             (g[0] < 1).double().mean().item(), rows[0, i], rows[3, i], rows[5, i], rows[2, i]))
    worst = max(worst, d.max().item())
print("worst", worst)
