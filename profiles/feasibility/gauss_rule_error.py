import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
S, T = 20, 0.00139
xs = T * ((np.arange(1, S + 1) - 0.5) / S - 0.5)
H = xs.max()
def gauss_discrete(n, pts):
    # Gauss rule for the measure (1/S) sum delta(x - pts): Stieltjes + Golub-Welsch
    w = np.full(pts.size, 1.0 / pts.size)
    alpha, beta = np.zeros(n), np.zeros(n)
    p_prev, p = np.zeros_like(pts), np.ones_like(pts)
    norm_prev = 1.0
    for k in range(n):
        norm = np.sum(w * p * p)
        alpha[k] = np.sum(w * pts * p * p) / norm
        beta[k] = norm / norm_prev if k > 0 else norm
        p_next = (pts - alpha[k]) * p - (beta[k] if k > 0 else 0.0) * p_prev
        p_prev, p, norm_prev = p, p_next, norm
    J = np.diag(alpha) + np.diag(np.sqrt(beta[1:]), 1) + np.diag(np.sqrt(beta[1:]), -1)
    ev, V = np.linalg.eigh(J)
    return ev, beta[0] * V[0] ** 2
rng = np.random.default_rng(0)
rows = []
for _ in range(300):
    k = rng.choice([rng.uniform(0.02, 0.2), rng.uniform(0.2, 0.9)])
    a = rng.uniform(3, 30); b = rng.uniform(0, 1 + k); inc = np.arccos(b / a)
    rows.append((k, 0.0, rng.uniform(1, 30), a, inc, 0.0, 0.0))
rows = np.array(rows); u = np.array([[0.4, 0.25]] * len(rows))
res = {n: [] for n in (6, 7, 8, 9)}
for r, ld in zip(rows, u):
    k, _, p, a, inc, _, _ = r
    v = 2 * np.pi * a / p; b = a * np.cos(inc)
    tc = np.linspace(-1.3 * (1 + k) / v, 1.3 * (1 + k) / v, 801)
    ref = O.evaluate_pv(tc, r[None, :], ld[None, :], T, S)[0]
    D = np.full(tc.size, np.inf)
    for c in (1 + k, abs(1 - k)):
        s2 = (c * c - b * b) / (a * a - b * b)
        root = np.arcsin(np.sqrt(complex(s2))) * p / (2 * np.pi)
        for s in (root, -root):
            D = np.minimum(D, np.abs(tc - s) / H)
    for n in res:
        x, w = gauss_discrete(n, xs)
        tt = (tc[:, None] + x[None, :]).ravel()
        f = O.evaluate_pv(tt, r[None, :], ld[None, :], 0.0, 1)[0].reshape(tc.size, n)
        res[n].append(np.stack([D, np.abs(1 - (1 - f) @ w - ref)], 1))
for n in res:
    A = np.concatenate(res[n])
    print("n=%2d" % n, end="  ")
    for lo, hi in ((1.2, 1.5), (1.5, 1.8), (1.8, 2.0), (2.0, 2.3), (2.3, 2.8), (2.8, 3.5), (3.5, 5)):
        m = (A[:, 0] >= lo) & (A[:, 0] < hi)
        print("D[%g,%g) %.1e" % (lo, hi, A[m, 1].max() if m.any() else 0), end=" | ")
    print()
x, w = gauss_discrete(3, xs / T); print(x, w, w.sum())
