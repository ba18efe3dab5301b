"""Prior samplers and companion / background priors of the marginal-likelihood path.

Mirrors the call surface of the reference's triceratops/priors.py (same names, argument
meaning and return values) for the functions on the hot path:
  sample_rp (priors.py:16-116), sample_inc (119-132), sample_ecc (134-155), sample_w (157-166),
  sample_q (168-274), sample_q_companion (277-383),
  lnprior_bound_TP (580-782), lnprior_bound_EB (784-984), lnprior_background (986-1005).
The dead priors (priors.py:386-577: no caller anywhere in the reference) are out of scope.

The broken power laws are expressed once, as a table of segments fed to one inverse-CDF
routine, instead of one hand-expanded branch per mass range; floating-point operations are
kept in the reference's order so that the same uniforms give the same draws bit for bit.
Unlike the reference the samplers never modify the caller's array.
"""
import numpy as np

from .constants import G, Msun, au, pi
from .funcs import separation_at_contrast


# ---------------------------------------------------------------------------------------
# piecewise power-law inverse CDF
def _broken_power_law(edges, powers, amps):
    """Segments of a continuous-or-stepped broken power law.

    edges: n+1 break points, powers: n exponents, amps: n amplitude factors (the first is 1).
    Returns (norm, cumulative integrals, segments) in the arithmetic the reference uses:
    I_j = amp_j * (hi^(p+1) - lo^(p+1)) / (p+1),  Norm = 1/sum(I).
    """
    ints = []
    for j, p in enumerate(powers):
        span = edges[j + 1] ** (p + 1) - edges[j] ** (p + 1)
        ints.append(span / (p + 1) if amps[j] is None else amps[j] * span / (p + 1))
    total = ints[0]
    for v in ints[1:]:
        total = total + v
    return 1 / total, ints


def _invert(x, norm, ints, edges, powers, amps, select=None):
    """Inverse CDF of the law above evaluated at uniforms x (values past the last edge of
    the CDF are returned unchanged, like the reference's masked assignment)."""
    out = np.array(x, dtype=np.float64, copy=True)
    cum = None
    for j, p in enumerate(powers):
        upper = ints[j] if cum is None else cum + ints[j]
        if cum is None:
            m = x <= norm * upper
        else:
            m = (x > norm * cum) & (x <= norm * upper)
        if select is not None:
            m = m & select
        t = x[m] / norm
        for prev in ints[:j]:
            t = t - prev
        t = t * (p + 1)
        if amps[j] is not None:
            t = t / amps[j]
        out[m] = (t + edges[j] ** (p + 1)) ** (1 / (p + 1))
        cum = upper
    return out


def sample_rp(x, M_s, flatpriors):
    """Planet radii [Earth radii] from uniforms x; host masses M_s (array) select the law.
    Reference: priors.py:16-116 (breaks at 3 and 6 R_earth, range 0.5-20, split at 0.45 M_sun)."""
    x = np.asarray(x, dtype=np.float64)
    if flatpriors == True:  # noqa: E712  (reference compares with ==)
        A = 1 / 19.5
        return x / A + 0.5
    M_s = np.asarray(M_s, dtype=np.float64)
    edges = (0.5, 3.0, 6.0, 20.0)
    out = np.array(x, copy=True)
    for powers, sel in (((0.0, -4.0, -0.5), M_s > 0.45), ((0.0, -7.0, -0.5), M_s <= 0.45)):
        p1, p2, p3 = powers
        A1 = edges[1] ** p1 / edges[1] ** p2
        A2 = edges[2] ** p2 / edges[2] ** p3
        amps = (None, A1, A2 * A1)
        norm, ints = _broken_power_law(edges, powers, amps)
        amps_inv = (None, A1, A1 * A2)
        part = _invert(x, norm, ints, edges, powers, amps_inv, select=sel)
        out[sel] = part[sel]
    return out


def sample_inc(x, lower=0, upper=90):
    """Inclinations [deg], uniform in cos i (priors.py:119-132)."""
    Norm = 1 / (np.cos(lower * np.pi / 180) - np.cos(upper * np.pi / 180))
    return np.arccos(np.cos(lower * np.pi / 180) - x / Norm) * 180 / np.pi


def sample_ecc(x, planet, P_orb):
    """Eccentricities (priors.py:134-155).  As in the reference the uniforms x only fix the
    sample size: the draw comes from the global numpy stream -- Beta(0.867, 3.030) for planets,
    a power law of index 0.2 (P_orb <= 10 d) or 0.6 for binaries.  scipy.stats.beta.rvs /
    powerlaw.rvs consume the global stream exactly like the two numpy calls below."""
    size = len(x)
    if planet == True:  # noqa: E712
        return np.random.beta(0.867, 3.030, size=size)
    a = 0.2 if P_orb <= 10 else 0.6
    return np.random.uniform(size=size) ** (1.0 / a)


def sample_w(x):
    """Arguments of periastron [deg] (priors.py:157-166)."""
    return x * 360


def _mass_ratio(x, M_s, p_hi, F_twin):
    """Shared body of sample_q / sample_q_companion (priors.py:168-383): power 0.3 below
    q = 0.3, power p_hi above, an excess twin fraction F_twin in [0.95, 1], lower limit
    q_min = 0.1/M_s for M_s < 1."""
    x = np.asarray(x, dtype=np.float64)
    if M_s <= 0.1:
        return np.full(len(x), 1.0)
    p1, p2 = 0.3, p_hi

    def twin_amp(lo):
        return (1 + (F_twin) / (1 - F_twin)
                * ((1.0 ** (p2 + 1) - lo ** (p2 + 1)) / (p2 + 1))
                / ((1.0 ** (p2 + 1) - 0.95 ** (p2 + 1)) / (p2 + 1)))

    if M_s >= 0.3:
        q_min = 0.1 if M_s >= 1.0 else 0.1 / M_s
        A1 = (0.3 ** p1) / (0.3 ** p2)
        A2 = twin_amp(0.3)
        edges = (q_min, 0.3, 0.95, 1.0)
        powers = (p1, p2, p2)
        norm, ints = _broken_power_law(edges, powers, (None, A1, A2 * A1))
        return _invert(x, norm, ints, edges, powers, (None, A1, A1 * A2))
    q_min = 0.1 / M_s
    A2 = twin_amp(q_min)
    edges = (q_min, 0.95, 1.0)
    powers = (p2, p2)
    norm, ints = _broken_power_law(edges, powers, (None, A2))
    return _invert(x, norm, ints, edges, powers, (None, A2))


def sample_q(x, M_s):
    """Mass ratios of short-period binaries (priors.py:168-274)."""
    return _mass_ratio(x, M_s, -0.5, 0.30)


def sample_q_companion(x, M_s):
    """Mass ratios of long-period companions (priors.py:277-383)."""
    return _mass_ratio(x, M_s, -0.95, 0.05)


# ---------------------------------------------------------------------------------------
# bound-companion rate (Moe & Di Stefano 2017 parametrisation used by the reference)
def _bound_rate(M_s, plx, delta_mags, separations, contrasts, keep_close):
    """f_comp of priors.py:580-984.  keep_close=False is the TP flavour (companions with
    log10 P < 3.4 are not counted, :660-674), True the EB flavour (:861-876)."""
    if np.isnan(plx):
        plx = 0.1
    d = 1000 / plx
    seps = d * separation_at_contrast(delta_mags, separations, contrasts)
    M_act = M_s
    M_ref = M_s if M_s >= 1.0 else 1.0
    lm = np.log10(M_ref)
    f1 = 0.020 + 0.04 * lm + 0.07 * (lm) ** 2
    f2 = 0.039 + 0.07 * lm + 0.01 * (lm) ** 2
    f3 = 0.078 - 0.05 * lm + 0.04 * (lm) ** 2
    alpha = 0.018
    dlogP = 0.7
    max_Porbs = ((4 * pi ** 2) / (G * M_ref * Msun) * (seps * au) ** 3) ** (1 / 2) / 86400
    lp = np.log10(max_Porbs)

    t2_partial = 0.5 * (lp - 1.0) * (2.0 * f1 + (f2 - f1 - alpha * dlogP) * (lp - 1.0))
    t2 = 0.5 * (2.0 - 1.0) * (2.0 * f1 + (f2 - f1 - alpha * dlogP) * (2.0 - 1.0))
    t3_partial = 0.5 * alpha * (lp ** 2 - 5.4 * lp + 6.8) + f2 * (lp - 2.0)
    t3 = 0.5 * alpha * (3.4 ** 2 - 5.4 * 3.4 + 6.8) + f2 * (3.4 - 2.0)
    t4_partial = (alpha * dlogP * (lp - 3.4) + f2 * (lp - 3.4)
                  + (f3 - f2 - alpha * dlogP) * (0.238095 * lp ** 2 - 0.952381 * lp + 0.485714))
    t4 = (alpha * dlogP * (5.5 - 3.4) + f2 * (5.5 - 3.4)
          + (f3 - f2 - alpha * dlogP) * (0.238095 * 5.5 ** 2 - 0.952381 * 5.5 + 0.485714))
    t5_partial = f3 * (3.33333 - 17.3566 * np.exp(-0.3 * lp))
    t5 = f3 * (3.33333 - 17.3566 * np.exp(-0.3 * 8.0))

    f_comp = np.zeros(len(seps))
    b1 = (lp >= 1.0) & (lp < 2.0)
    b2 = (lp >= 2.0) & (lp < 3.4)
    b3 = (lp >= 3.4) & (lp < 5.5)
    b4 = (lp >= 5.5) & (lp < 8.0)
    b5 = lp >= 8.0
    if keep_close:
        f_comp[b1] = t2_partial[b1]
        f_comp[b2] = t2 + t3_partial[b2]
        f_comp[b3] = t2 + t3 + t4_partial[b3]
        f_comp[b4] = t2 + t3 + t4 + t5_partial[b4]
        f_comp[b5] = t2 + t3 + t4 + t5
    else:
        f_comp[b3] = t4_partial[b3]
        f_comp[b4] = t4 + t5_partial[b4]
        f_comp[b5] = t4 + t5
    if M_s >= 1.0:
        return np.log(f_comp)
    f_act = 0.65 * f_comp + 0.35 * f_comp * M_act
    f_act[f_act < 0.0] = 0.0
    return np.log(f_act)


def lnprior_bound_TP(M_s: float, plx: float, delta_mags, separations, contrasts):
    """Log bound-companion rate for planet scenarios (priors.py:580-782)."""
    with np.errstate(divide="ignore"):
        return _bound_rate(M_s, plx, delta_mags, separations, contrasts, keep_close=False)


def lnprior_bound_EB(M_s: float, plx: float, delta_mags, separations, contrasts):
    """Log bound-companion rate for EB scenarios (priors.py:784-984)."""
    with np.errstate(divide="ignore"):
        return _bound_rate(M_s, plx, delta_mags, separations, contrasts, keep_close=True)


def lnprior_background(N_comp: int, delta_mags, separations, contrasts):
    """Log probability of a chance-aligned background star inside the limiting separation
    (priors.py:986-1005): natural log of (N_comp/0.1) (1/3600)^2 sep^2."""
    seps = separation_at_contrast(delta_mags, separations, contrasts)
    with np.errstate(divide="ignore"):
        return np.log((N_comp / 0.1) * (1 / 3600) ** 2 * seps ** 2)
