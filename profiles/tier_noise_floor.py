"""Is the ~3e-13 worst case of profiles/tier_stress.py quadrature error or rounding noise of extreme orbits?
For the rows with the largest on/off difference: both GPU variants against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from triceratops_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
exptime, S, span, npts = 0.00139, 20, 0.5, 3000
n = 8192
k = 10 ** rng.uniform(np.log10(0.003), np.log10(1.6), n)
a = 10 ** rng.uniform(np.log10(1.3), np.log10(300), n)
e = np.where(rng.random(n) < 0.4, 0.0, rng.uniform(0, 0.97, n))
w = rng.uniform(0, 2 * np.pi, n)
b = rng.uniform(0, 1 + k) * np.where(rng.random(n) < 0.2, 1.0, rng.uniform(0.9, 1.0, n))
inc = np.arccos(np.clip(b / (a * (1 - e * e) / (1 + e * np.sin(w))), 0, 1))
per = 10 ** rng.uniform(np.log10(0.25), 3, n)
ok = (a * (1 - e) > 1 + k) & (k <= 1.0)
rows = np.ascontiguousarray(np.stack([k, rng.uniform(-0.05, 0.05, n), per, a, inc, e, w,
                                      rng.uniform(0.0, 0.8, n), rng.uniform(-0.1, 0.5, n)])[:, ok])
t = np.sort(rng.uniform(-span, span, npts))
g = {}
for on in (1, 0):
    L.trx_set_supersample_tiers(on)
    g[on] = _lib.flux_grid(_lib.MODEL_RAW, 0, _lib.dev(t), _lib.dev(rows), exptime, S, False)[0].cpu().numpy()
L.trx_set_supersample_tiers(1)
d = np.abs(g[1] - g[0]).max(axis=1)
top = np.argsort(-d)[:12]
want = O.evaluate_pv(t, rows[:7, top].T, rows[7:, top].T, exptime, S)
for j, r in enumerate(top):
    print("k=%.3f a=%7.2f e=%.2f P=%7.2f  |on-off| %.2e   |on-oracle| %.2e   |off-oracle| %.2e"
          % (rows[0, r], rows[3, r], rows[5, r], rows[2, r], d[r], np.abs(g[1][r] - want[j]).max(), np.abs(g[0][r] - want[j]).max()))
