import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from numpy.polynomial import chebyshev as Ch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gauss_rule_error.py')).read().split("rng = np.random.default_rng(0)")[0])   # S, T, xs, H, gauss_discrete
m = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Dc = float(sys.argv[2]) if len(sys.argv) > 2 else 1.7
nq = 5
xq, wq = gauss_discrete(nq, xs)
rng = np.random.default_rng(1)
t = np.linspace(-0.25, 0.25, 2000)
errs, cov, inwin = [], 0, 0
theta = np.pi * (2 * np.arange(m) + 1) / (2 * m)
xn = np.cos(theta)
for _ in range(150):
    k = rng.choice([rng.uniform(0.02, 0.2), rng.uniform(0.2, 0.9)])
    a = rng.uniform(3, 30); b = rng.uniform(0, 1 + k); inc = np.arccos(b / a); p = rng.uniform(1, 30)
    r = np.array([k, 0.0, p, a, inc, 0.0, 0.0]); ld = np.array([0.4, 0.25])
    ref = O.evaluate_pv(t, r[None, :], ld[None, :], T, S)[0]
    v = 2 * np.pi * a / p
    roots = []
    for c in (1 + k, abs(1 - k)):
        s2 = (c * c - b * b) / (a * a - b * b)
        rt = np.arcsin(np.sqrt(complex(s2))) * p / (2 * np.pi)
        roots += [rt, -rt]
    roots = np.array(roots)
    kscale = 1 + 0.5 * min(k, 1) ** 2
    inwin += int((ref < 1).sum())
    for j0 in range(0, 2000, 64):
        tj = t[j0:j0 + 64]
        tc, W = 0.5 * (tj.min() + tj.max()), 0.5 * (tj.max() - tj.min()) + 0.5 * T
        R = np.min(np.abs(tc - roots))
        occ = O.evaluate_pv(np.array([tc]), r[None, :], ld[None, :], 0.0, 1)[0][0] < 1
        if not (occ and R >= max(Dc * W, W - H + 6.5 * H) * kscale):
            continue
        fn = O.evaluate_pv(tc + W * xn, r[None, :], ld[None, :], 0.0, 1)[0]
        coef = Ch.chebfit(xn, 1 - fn, m - 1)                      # interpolation through the m nodes
        tt = (tj[:, None] + xq[None, :] - tc) / W
        est = 1 - (Ch.chebval(tt, coef) @ wq)
        errs.append(np.abs(est - ref[j0:j0 + 64]).max())
        cov += tj.size
print("m=%d Dc=%.2f: chunks %d  max err %.2e  p99 %.2e   covered %.3f of occulted cells" % (m, Dc, len(errs), max(errs), np.quantile(errs, 0.99), cov / inwin))
