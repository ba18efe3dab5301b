"""Synthetic parameter blocks and light curves for tests and bench.py.

Host-side numpy only.  Every row is drawn so that it transits (inc >= inc_min), therefore
evaluations = n_time * n_rows exactly.
"""
import numpy as np

from .constants import G, Msun, Rearth, Rsun
from ._lib import MODEL_EB, MODEL_EB_TWIN, MODEL_TP

SEED = 20260424
EXPTIME = 0.00139
NSAMPLES = 20
SIGMA = 5e-4

# the 18 scenario families of calc_probs (triceratops.py:673-1485): (name, model, is_host, has_companion)
FAMILIES = []
for _pre, _host, _comp in (("T", False, False), ("P", False, True), ("S", True, True),
                           ("D", False, True), ("B", True, True), ("N", False, False)):
    FAMILIES.append((_pre + "TP", MODEL_TP, _host, _comp))
    FAMILIES.append((_pre + "EB", MODEL_EB, _host, _comp))
    FAMILIES.append((_pre + "EBx2P", MODEL_EB_TWIN, _host, _comp))


def time_grid(n_time):
    return np.linspace(-0.25, 0.25, n_time)


def reference_tp_row():
    """k=0.07, a/R=12, i=88.5 deg, e=0, u=(0.4,0.25), P=3 d, R_s=1."""
    R_s = 1.0
    return np.array([[0.07 * Rsun / Rearth], [3.0], [88.5], [12.0 * R_s * Rsun], [R_s], [0.4],
                     [0.25], [0.0], [0.0], [0.0]])


def _orbit(rng, n, R_x_cm, R_s, M_tot, ecc, period_scale=1.0):
    P = rng.uniform(1.0, 30.0, n) * period_scale
    a = ((G * M_tot * Msun) / (4 * np.pi ** 2) * (P * 86400) ** 2) ** (1 / 3)
    argp = rng.uniform(0.0, 360.0, n)
    e_corr = (1 + ecc * np.sin(argp * np.pi / 180)) / (1 - ecc ** 2)
    Ptra = np.clip((R_x_cm + R_s * Rsun) / a * e_corr, 0.0, 1.0)
    inc_min = np.arccos(Ptra) * 180 / np.pi
    inc = rng.uniform(inc_min, 90.0)
    return P, a, inc, argp


def tp_rows(rng, n, has_companion=False):
    """TP parameter block [10][n]: R_p P inc a R_s u1 u2 ecc argp comp_fr."""
    R_p = rng.uniform(0.5, 20.0, n)
    R_s = rng.uniform(0.3, 2.0, n)
    M_s = rng.uniform(0.3, 2.0, n)
    ecc = np.minimum(rng.beta(0.867, 3.03, n), 0.9)
    P, a, inc, argp = _orbit(rng, n, R_p * Rearth, R_s, M_s, ecc)
    u1 = rng.uniform(0.1, 0.6, n)
    u2 = rng.uniform(0.05, 0.4, n)
    fr = rng.uniform(0.01, 0.5, n) if has_companion else np.zeros(n)
    return np.ascontiguousarray(np.stack([R_p, P, inc, a, R_s, u1, u2, ecc, argp, fr]))


def eb_rows(rng, n, twin=False, has_companion=False):
    """EB parameter block [11][n]: R_EB EB_fr P inc a R_s u1 u2 ecc argp comp_fr."""
    R_s = rng.uniform(0.3, 2.0, n)
    M_s = rng.uniform(0.3, 2.0, n)
    if twin:
        R_EB = R_s.copy()
        q = rng.uniform(0.95, 1.0, n)
        EB_fr = rng.uniform(0.4, 0.5, n)
    else:
        R_EB = rng.uniform(0.1, 1.0, n) * R_s
        q = rng.uniform(0.1, 0.95, n)
        # log-uniform so that both sides of the 1.5-sigma secondary-depth rule are populated
        EB_fr = 10.0 ** rng.uniform(-5.0, np.log10(0.5), n)
    ecc = np.minimum(rng.uniform(0, 1, n) ** (1 / 0.2), 0.9)
    P, a, inc, argp = _orbit(rng, n, R_EB * Rsun, R_s, M_s * (1 + q), ecc,
                             period_scale=2.0 if twin else 1.0)
    u1 = rng.uniform(0.1, 0.6, n)
    u2 = rng.uniform(0.05, 0.4, n)
    fr = rng.uniform(0.01, 0.5, n) if has_companion else np.zeros(n)
    return np.ascontiguousarray(np.stack([R_EB, EB_fr, P, inc, a, R_s, u1, u2, ecc, argp, fr]))


def family_rows(rng, family, n):
    name, model, is_host, has_comp = family
    if model == MODEL_TP:
        return tp_rows(rng, n, has_comp)
    return eb_rows(rng, n, twin=(model == MODEL_EB_TWIN), has_companion=has_comp)


def noisy_light_curve(rng, model_curve, sigma=SIGMA):
    return model_curve + rng.normal(0.0, sigma, model_curve.shape)


def toi_jobs(n_tois, n_time=200, N=1_000_000, seed=SEED, trilegal_fname=None,
             contrast_curve_file=None, parallel=True):
    """BASELINE config 4: a batch of synthetic TOIs for calc_probs_many.  Each TOI is a K/G dwarf
    with a transiting planet, observed as an `n_time`-point folded light curve, plus ONE nearby star
    bright enough to host the signal: 15 + 3 = 18 scenarios per TOI.  Returns [(target, kwargs)].
    The light curves are evaluated on the GPU (likelihoods.simulate_TP_transit)."""
    from pandas import DataFrame
    from .likelihoods import simulate_TP_transit
    from .triceratops import target
    rng = np.random.default_rng(seed)
    jobs = []
    for i in range(n_tois):
        M_s = rng.uniform(0.6, 1.2)
        R_s = M_s ** 0.8
        Teff = 3600.0 + 2300.0 * M_s
        P = rng.uniform(2.0, 10.0)
        k = rng.uniform(0.04, 0.12)
        a = ((G * M_s * Msun) / (4 * np.pi ** 2) * (P * 86400) ** 2) ** (1 / 3)
        b = rng.uniform(0.0, 0.6)
        inc = np.degrees(np.arccos(b * R_s * Rsun / a))
        sigma = rng.uniform(3e-4, 8e-4)
        t = np.linspace(-0.2, 0.2, n_time)
        fr_nearby = rng.uniform(0.02, 0.08)
        curve = simulate_TP_transit(t, k * R_s * Rsun / Rearth, P, inc, a, R_s, 0.45, 0.2, 0.0, 0.0,
                                    companion_fluxratio=fr_nearby)
        flux = curve + rng.normal(0.0, sigma, n_time)
        depth = float(1.0 - curve.min())
        Tmag = rng.uniform(9.5, 12.0)
        stars = DataFrame({
            "ID": [1000 + 2 * i, 1001 + 2 * i], "Tmag": [Tmag, Tmag + 3.0],
            "Jmag": [Tmag - 0.8, Tmag + 2.1], "Hmag": [Tmag - 1.2, Tmag + 1.6],
            "Kmag": [Tmag - 1.3, Tmag + 1.5], "ra": [10.0, 10.004], "dec": [-5.0, -5.003],
            "mass": [M_s, rng.uniform(0.4, 0.9)], "rad": [R_s, rng.uniform(0.4, 0.9)],
            "Teff": [Teff, rng.uniform(3600.0, 5400.0)], "plx": [rng.uniform(5.0, 20.0), rng.uniform(1.0, 4.0)],
            "fluxratio": [1.0 - fr_nearby, fr_nearby],
            "tdepth": [depth / (1.0 - fr_nearby), min(depth / fr_nearby, 0.95)]})
        tg = target(1000 + 2 * i, np.array([1]), stars=stars, trilegal_fname=trilegal_fname)
        jobs.append((tg, dict(time=t, flux_0=flux, flux_err_0=sigma, P_orb=float(P),
                              contrast_curve_file=contrast_curve_file, N=N, parallel=parallel)))
    return jobs
