import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
S, T = 20, 0.00139
xs = T * ((np.arange(1, S + 1) - 0.5) / S - 0.5)          # sub-exposure offsets
H = xs.max()
def cheb_weights(n):
    x = H * np.cos((2 * np.arange(n) + 1) * np.pi / (2 * n))      # Chebyshev-Gauss nodes on [-H, H]
    w = np.zeros(n)
    for j in range(n):
        l = np.ones_like(xs)
        for m in range(n):
            if m != j:
                l *= (xs - x[m]) / (x[j] - x[m])
        w[j] = l.mean()
    return x, w
rng = np.random.default_rng(0)
rows = []
for _ in range(300):
    k = rng.choice([rng.uniform(0.02, 0.2), rng.uniform(0.2, 0.9)])
    a = rng.uniform(3, 30); b = rng.uniform(0, 1 + k); inc = np.arccos(b / a)
    p = rng.uniform(1, 30)
    rows.append((k, 0.0, p, a, inc, 0.0, 0.0))
rows = np.array(rows)
u = np.array([[0.4, 0.25]] * len(rows))
res = {n: [] for n in (4, 5, 6, 7, 8, 10, 12)}
for r, ld in zip(rows, u):
    k, _, p, a, inc, _, _ = r
    v = 2 * np.pi * a / p                                    # sky speed, stellar radii per day
    b = a * np.cos(inc)
    tc = np.linspace(-1.3 * (1 + k) / v, 1.3 * (1 + k) / v, 801)
    ref = O.evaluate_pv(tc, r[None, :], ld[None, :], T, S)[0]
    # distance (in units of H) from the exposure centre to the nearest singular time, real or complex:
    # z^2 = (v t)^2 + b^2 (straight-line approx, fine for the study); roots of z^2 = c^2
    D = np.full(tc.size, np.inf)
    for c in (1 + k, abs(1 - k), k):
        s2 = (c * c - b * b) / (a * a - b * b)
        root = np.arcsin(np.sqrt(complex(s2))) * p / (2 * np.pi)
        for s in (root, -root):
            D = np.minimum(D, np.abs(tc - s) / H)
    for n in res:
        x, w = cheb_weights(n)
        tt = (tc[:, None] + x[None, :]).ravel()
        f = O.evaluate_pv(tt, r[None, :], ld[None, :], 0.0, 1)[0].reshape(tc.size, n)
        res[n].append(np.stack([D, np.abs(f @ w - ref), np.full(tc.size, k)], 1))
for n in res:
    A = np.concatenate(res[n])
    print("n=%2d" % n, end="  ")
    for lo, hi in ((1.5, 2), (2, 3), (3, 4), (4, 6), (6, 8), (8, 12), (12, 20), (20, 40), (40, 1e9)):
        m = (A[:, 0] >= lo) & (A[:, 0] < hi)
        print("D[%g,%g) %.1e" % (lo, hi, A[m, 1].max() if m.any() else 0), end=" | ")
    print()
