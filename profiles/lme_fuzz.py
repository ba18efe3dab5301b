"""log-mean-exp against the oracle over sizes around the segment / block marks, -inf / NaN mixes, aligned and
misaligned buffers, and the fused form with a prior: python profiles/lme_fuzz.py  (last run: worst |error| 3.6e-15)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from triceratops_amd import _lib
rng = np.random.default_rng(0)
worst = 0.0
sizes = [1, 2, 3, 511, 512, 513, 4095, 4096, 4097, 2048*512-1, 2048*512, 2048*512+1, 2048*512*2+5, 999_983, 3_000_017]
for n in sizes:
    for kind in ("wide", "narrow", "inf", "nan"):
        x = rng.uniform(-3000, -1, n) if kind != "narrow" else rng.uniform(-30, -1, n)
        if kind == "inf": x[rng.random(n) < 0.7] = -np.inf
        if kind == "nan": x[rng.random(n) < 0.1] = np.nan
        want = O.log_mean_exp(x, n)
        for off in (0, 1):           # aligned / 8-byte-misaligned start (scalar path)
            buf = torch.empty(n + 1, dtype=torch.float64, device="cuda")
            buf[off:off + n] = torch.as_tensor(x).cuda()
            got = float(_lib.log_mean_exp(buf[off:off + n], n).cpu()[0])
            err = abs(got - want) if np.isfinite(want) else (0.0 if got == want else 1.0)
            worst = max(worst, err)
            if err > 1e-11: print("FAIL", n, kind, off, got, want)
        # fused form with a prior
        pr = rng.uniform(-5, 0, n)
        lnz = float(_lib.lnz_from_halfchi2(_lib.dev(-x.copy()), _lib.dev(pr), n + 7, 0.3).cpu()[0])
        lw = np.full(n + 7, -np.inf); lw[:n] = -0.5 * np.log(2 * np.pi) - 0.3 - (-x) + pr
        w2 = O.log_mean_exp(lw, n + 7)
        e2 = abs(lnz - w2) if np.isfinite(w2) else (0.0 if (lnz == w2 or (np.isnan(lnz) and np.isnan(w2))) else 1.0)
        worst = max(worst, e2)
        if e2 > 1e-10: print("FAIL fused", n, kind, lnz, w2)
print("worst", worst)
