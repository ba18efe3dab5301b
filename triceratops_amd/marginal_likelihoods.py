"""Marginal likelihoods (evidences) of the triceratops scenarios on the MI355X kernels.

Same entry points, positional signatures and return conventions as the reference's
triceratops/marginal_likelihoods.py for the ten functions calc_probs calls:
  lnZ_TTP (39-172), lnZ_TEB (175-383), lnZ_PTP (386-586), lnZ_PEB (589-866), lnZ_STP (869-1077),
  lnZ_SEB (1080-1376), lnZ_DTP (1379-1568), lnZ_DEB (1571-1837), lnZ_BTP (1840-2035),
  lnZ_BEB (2038-2362),
and for the four it only exports: lnZ_NTP_unknown (2365), lnZ_NEB_unknown (2554),
lnZ_NTP_evolved (2832), lnZ_NEB_evolved (2969).
TP family -> one dict, EB family -> (res, res_twin); every dict holds the 14 best-fit columns
(100 draws, by decreasing lnL) and 'lnZ' (Python float, may be -inf).

How one call runs:
  host (numpy)  draw the priors from the *global* numpy stream in the reference's order, derive
                the per-draw stellar/orbital columns, build the geometry mask and lnprior_companion
  device (HIP)  the n masked draws are packed into one SoA block; trx_lnz_scenario evaluates the
                supersampled light-curve model, chi^2/2 per draw and the log-mean-exp evidence
  host          lnL vector, best-100 table
`parallel=True` reproduces the reference's vector path, `parallel=False` its per-draw loop
semantics (draws with Ptra > 1 skipped, scalar radius-ratio rule; App. C of SURVEY.md) -- both
run on the GPU, there is no Python loop over draws here.
"""
from pathlib import Path

import numpy as np
import torch
from pandas import read_csv

from . import _lib
from ._lib import FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K, MODEL_EB, MODEL_EB_TWIN, MODEL_TP
from ._numerics import _log_mean_exp  # noqa: F401  (same module surface as the reference)
from .constants import G, Msun, Rearth, Rsun, au, ln2pi, pi  # noqa: F401
from .funcs import (file_to_contrast_curve, flux_relation, stellar_relations, trilegal_results)
from .likelihoods import *  # noqa: F401,F403  (the reference re-exports these names)
from .priors import *  # noqa: F401,F403
from .priors import (lnprior_background, lnprior_bound_EB, lnprior_bound_TP, sample_ecc, sample_inc,
                     sample_q, sample_q_companion, sample_rp, sample_w)

np.seterr(divide='ignore')

_DATA_DIR = Path(__file__).parent / "data"
N_BEST = 100


# ---------------------------------------------------------------------------------------
# limb-darkening tables (Claret grids shipped as data; marginal_likelihoods.py:21-37)
class _LdcTable:
    def __init__(self, fname, c1, c2):
        df = read_csv(_DATA_DIR / fname)
        self.Zs = np.array(df.Z, dtype=float)
        self.Teffs = np.array(df.Teff, dtype=int)
        self.loggs = np.array(df.logg, dtype=float)
        self.u1s = np.array(df[c1], dtype=float)
        self.u2s = np.array(df[c2], dtype=float)

    def _one(self, sel):
        """exactly-one-row lookup; raises ValueError like ndarray.item() on size != 1"""
        return self.u1s[sel].item(), self.u2s[sel].item()

    def star(self, Z, Teff, logg):
        """nearest node in Z, Teff and logg independently (e.g. marginal_likelihoods.py:90-98)"""
        z = self.Zs[np.argmin(np.abs(self.Zs - Z))]
        t = self.Teffs[np.argmin(np.abs(self.Teffs - Teff))]
        g = self.loggs[np.argmin(np.abs(self.loggs - logg))]
        return self._one((self.Zs == z) & (self.Teffs == t) & (self.loggs == g))

    def companions(self, Z, Teffs, loggs, teff_cap):
        """per-draw coefficients on the rounded (Teff/250, logg/0.5) grid at the nearest Z
        (marginal_likelihoods.py:945-972; cap 10000 K for STP, 13000 K for SEB :1181)."""
        atZ = self.Zs == self.Zs[np.abs(self.Zs - Z).argmin()]
        tz, gz, a1, a2 = self.Teffs[atZ], self.loggs[atZ], self.u1s[atZ], self.u2s[atZ]
        rg = np.round(loggs / 0.5) * 0.5
        rg[rg < 3.5] = 3.5
        rg[rg > 5.0] = 5.0
        rt = np.round(Teffs / 250) * 250
        rt[rt < 3500] = 3500
        rt[rt > teff_cap] = teff_cap
        # the rounded values live on a 250 K x 0.5 dex lattice: code them as integers and look
        # each distinct cell up once (an absent cell raises like the reference's .item())
        code = np.rint((rt - 3500) / 250).astype(np.int64) * 8 + np.rint((rg - 3.5) / 0.5).astype(np.int64)
        cells = np.unique(code)
        lut1, lut2 = np.full(cells.max() + 1, np.nan), np.full(cells.max() + 1, np.nan)
        for cidx in cells:
            sel = (tz == 3500 + 250 * (cidx // 8)) & (gz == 3.5 + 0.5 * (cidx % 8))
            lut1[cidx], lut2[cidx] = a1[sel].item(), a2[sel].item()
        return lut1[code], lut2[code]

    def field_stars(self, Teffs, loggs, Zs):
        """per-background-star coefficients: nearest Teff and logg, then the nearest Z among the
        rows of that (Teff, logg) (marginal_likelihoods.py:1913-1924).

        The reference compares every star with every table row; here the grid is indexed once --
        distinct Teff / logg nodes in their order of first appearance (np.argmin's tie rule picks the
        first row, i.e. the node that appears first) and, per (Teff, logg) cell, its rows in table
        order -- and a star looks at its cell's few rows only.  Same selections, same errors."""
        ix = self._index()
        it = np.argmin(np.abs(ix["uT"][None, :] - Teffs[:, None]), axis=1)
        ig = np.argmin(np.abs(ix["uG"][None, :] - loggs[:, None]), axis=1)
        c = ix["cell"][it, ig]
        if np.any(c < 0):                       # no row at that (Teff, logg): the reference's .item() fails
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        dz = np.abs(ix["Z"][c] - Zs[:, None])   # (stars, rows of the cell; padding = inf)
        slot = np.argmin(dz, axis=1)
        if not np.all(ix["same"][c, slot] == 1):
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        idx = ix["row"][c, slot]
        return self.u1s[idx], self.u2s[idx]

    def _index(self):
        ix = getattr(self, "_ix", None)
        if ix is not None:
            return ix

        def nodes(v):
            _, first = np.unique(v, return_index=True)
            u = v[np.sort(first)]
            return u, {x: i for i, x in enumerate(u.tolist())}

        uT, posT = nodes(self.Teffs)
        uG, posG = nodes(self.loggs)
        members = {}
        for j, (t, g) in enumerate(zip(self.Teffs.tolist(), self.loggs.tolist())):
            members.setdefault((posT[t], posG[g]), []).append(j)
        width = max(len(m) for m in members.values())
        cell = np.full((uT.size, uG.size), -1, dtype=np.int64)
        Z = np.full((len(members), width), np.inf)
        row = np.zeros((len(members), width), dtype=np.int64)
        same = np.zeros((len(members), width), dtype=np.int64)
        for c, ((a, b), m) in enumerate(members.items()):
            cell[a, b] = c
            z = self.Zs[m]
            Z[c, :len(m)], row[c, :len(m)] = z, m
            same[c, :len(m)] = (z[None, :] == z[:, None]).sum(axis=1)      # rows of the cell at this Z
        self._ix = {"uT": uT, "uG": uG, "cell": cell, "Z": Z, "row": row, "same": same}
        return self._ix


_tables = {}


def _ldc(mission):
    key = "TESS" if mission == "TESS" else "Kepler"
    if key not in _tables:
        _tables[key] = (_LdcTable("ldc_tess.csv", "aLSM", "bLSM") if key == "TESS"
                        else _LdcTable("ldc_kepler.csv", "a", "b"))
    return _tables[key]


# ---------------------------------------------------------------------------------------
# shared pieces of every scenario
def _periods(P_orb, N):
    """fixed period, or a uniform draw when a [lo, hi] range is given; numpy scalars take the
    range branch like the reference's `type(P_orb) not in [float, int]` test"""
    if type(P_orb) not in [float, int]:
        return np.random.uniform(low=P_orb[0], high=P_orb[-1], size=N)
    return np.full(N, P_orb)


def _sma(M_tot, P_days):
    """Kepler's third law, the reference's expression.  A fixed period with a scalar mass gives N
    identical values: the fractional power (0.27 s per 10^6 elements) is then taken on a short
    array -- the same vectorised code path as the full one, hence the same bits -- and broadcast."""
    if (np.ndim(M_tot) == 0 and np.ndim(P_days) == 1 and P_days.size > 16
            and P_days[0] == P_days[-1] and np.all(P_days == P_days[0])):
        head = ((G * M_tot * Msun) / (4 * pi ** 2) * (P_days[:16] * 86400) ** 2) ** (1 / 3)
        return np.full(P_days.size, head[0])
    return ((G * M_tot * Msun) / (4 * pi ** 2) * (P_days * 86400) ** 2) ** (1 / 3)


def _logg(M, R):
    return np.log10(G * (M * Msun) / (R * Rsun) ** 2)


def _flux_share(masses, M_s, filt="TESS"):
    """F(m) / (F(m) + F(M_s)) in band `filt` (the spline is evaluated once, the reference
    evaluates the same expression twice)"""
    f = flux_relation(masses, filt)
    return f / (f + flux_relation(np.array([M_s]), filt))


def _e_corr(eccs, argps):
    return (1 + eccs * np.sin(argps * pi / 180)) / (1 - eccs ** 2)


def _impact(a, eccs, argps, incs, R_host):
    r = a * (1 - eccs ** 2) / (1 + eccs * np.sin(argps * np.pi / 180))
    return r * np.cos(incs * pi / 180) / (R_host * Rsun)


def _transits(Ptra, incs, parallel):
    """draws inclined enough to transit.  Vector path: inc_min = 90 where Ptra > 1; serial path:
    such draws are skipped (`continue`)."""
    ok = Ptra <= 1.
    inc_min = np.full(len(Ptra), 90.)
    inc_min[ok] = np.arccos(Ptra[ok]) * 180. / pi
    hit = incs >= inc_min
    return hit if parallel else (hit & ok)


def _bound_companions(M_s, N, molusc_file):
    """mass ratios of unresolved bound companions: prior draw, or a MOLUSC table
    (marginal_likelihoods.py:455-464)"""
    if molusc_file is None:
        return sample_q_companion(np.random.rand(N), M_s)
    df = read_csv(molusc_file)
    sma = df["semi-major axis(AU)"].values
    ecc = df["eccentricity"].values
    qs = df[sma * (1 - ecc) > 10]["mass ratio"].values
    qs[qs < 0.1 / M_s] = 0.1 / M_s
    return np.pad(qs, (0, N - len(qs)))


def _clip_prior(lnprior, delta_mags):
    lnprior[lnprior > 0.0] = 0.0
    lnprior[delta_mags > 0.0] = -np.inf
    return lnprior


def _bound_prior(kind, M_s, plx, N, molusc_file, contrast_curve_file, fr_tess, fr_cc_fn):
    """lnprior_companion of the P and S scenarios (e.g. marginal_likelihoods.py:477-509).
    fr_tess: total companion flux term in the TESS band; fr_cc_fn(): same in the contrast-curve
    filter (evaluated only when a contrast curve is given)."""
    if molusc_file is not None:
        return np.zeros(N)
    fn = lnprior_bound_TP if kind == "TP" else lnprior_bound_EB
    if contrast_curve_file is None:
        delta_mags = 2.5 * np.log10(fr_tess)
        seps, cons = np.array([2.2]), np.array([1.0])
    else:
        delta_mags = 2.5 * np.log10(fr_cc_fn())
        seps, cons = file_to_contrast_curve(contrast_curve_file)
    return _clip_prior(fn(M_s, plx, np.abs(delta_mags), seps, cons), delta_mags)


class _Field:
    """TRILEGAL background population fainter than the target
    (marginal_likelihoods.py:1451-1461, 1887-1898, 2092-2106)"""

    def __init__(self, trilegal_fname, Tmag, Jmag, Hmag, Kmag):
        (self.Tmags, self.masses, self.loggs, self.Teffs, self.Zs, self.Jmags, self.Hmags,
         self.Kmags) = trilegal_results(trilegal_fname, Tmag)
        self.dT = Tmag - self.Tmags
        self.dJ = Jmag - self.Jmags
        self.dH = Hmag - self.Hmags
        self.dK = Kmag - self.Kmags
        self.fluxratios = 10 ** (self.dT / 2.5) / (1 + 10 ** (self.dT / 2.5))
        self.N_comp = self.Tmags.shape[0]

    def radii(self):
        return np.sqrt(G * self.masses * Msun / 10 ** self.loggs) / Rsun

    def band_delta(self, filt):
        return {"J": self.dJ, "H": self.dH, "K": self.dK}.get(filt, self.dT)

    def band_fluxratio(self, filt):
        d = self.band_delta(filt)
        return 10 ** (d / 2.5) / (1 + 10 ** (d / 2.5))

    def prior(self, N, idxs, contrast_curve_file, filt, fr_term=None, fr_term_cc=None):
        """lnprior_companion of the D and B scenarios (marginal_likelihoods.py:1465-1492,
        2161-2208).  fr_term / fr_term_cc: total flux terms (BEB adds the EB's share)."""
        if contrast_curve_file is None:
            if fr_term is None:
                fr_term = self.fluxratios[idxs] / (1 - self.fluxratios[idxs])
            delta_mags = 2.5 * np.log10(fr_term)
            lnprior = np.full(N, np.log((self.N_comp / 0.1) * (1 / 3600) ** 2 * 2.2 ** 2))
        else:
            if fr_term_cc is None:
                delta_mags = self.band_delta(filt)[idxs]
            else:
                delta_mags = 2.5 * np.log10(fr_term_cc)
            seps, cons = file_to_contrast_curve(contrast_curve_file)
            lnprior = lnprior_background(self.N_comp, np.abs(delta_mags), seps, cons)
        return _clip_prior(lnprior, delta_mags)


def _evidence(model, is_host, parallel, time, flux, sigma, cols, mask, lnprior, N, exptime,
              nsamples):
    """GPU part of a scenario branch: chi^2/2 of the masked draws and log-mean-exp over N.
    Returns (indices of the N_BEST best draws by decreasing lnL, lnZ)."""
    lnsigma = np.log(sigma)
    idx = np.flatnonzero(mask)
    block = np.empty((len(cols), idx.size), dtype=np.float64)
    for i, c in enumerate(cols):
        block[i] = c[idx] if isinstance(c, np.ndarray) else c
    flags = (FLAG_COMPANION_IS_HOST if is_host else 0) | (0 if parallel else FLAG_SCALAR_K)
    lp = None if lnprior is None else _lib.dev(lnprior[idx])
    h, lnz = _lib.lnz_scenario(model, flags, _lib.dev(time), _lib.dev(flux), sigma,
                               _lib.dev(block), exptime, nsamples, lp, N, lnsigma)
    # best draws = smallest chi^2/2.  With >= N_BEST finite values only their order matters and a
    # device top-k gives it; otherwise fall back to the reference's full argsort so that the
    # (arbitrary) order of the -inf ties is the reference's too.
    # (Equal chi^2 values do occur -- every draw whose model is flat over the data window has
    # the same one -- and their order is the sort algorithm's: any tie sends us to the fallback.)
    best = None
    if idx.size > N_BEST:
        hv, hi = torch.topk(h, N_BEST + 1, largest=False, sorted=True)
        if bool(torch.isfinite(hv[-1])) and bool((hv[1:] > hv[:-1]).all()):
            best = idx[hi[:N_BEST].cpu().numpy()]
    if best is None:
        lnL = np.full(N, -np.inf)
        lnL[idx] = -0.5 * ln2pi - lnsigma - h.cpu().numpy()
        best = (-lnL).argsort()[:N_BEST]
    return best, float(lnz.cpu()[0])


def _table(idx, lnZ, **cols):
    """best-N_BEST table, by decreasing lnL (marginal_likelihoods.py:152-171)"""
    res = {}
    for key in ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB",
                "R_EB", "fluxratio_EB", "fluxratio_comp"):
        v = cols[key]
        if isinstance(v, np.ndarray):
            res[key] = v[idx]
        else:
            res[key] = np.zeros(N_BEST) if v is None else np.full(N_BEST, v)
    res["lnZ"] = lnZ
    return res


def _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs, eccs,
                   argps, a, M_host, R_host, u1, u2, fr_comp, is_host, extra, lnprior):
    """common tail of the five *TP scenarios"""
    size = rps * Rearth + R_host * Rsun
    Ptra = size / a * _e_corr(eccs, argps)
    b = _impact(a, eccs, argps, incs, R_host)
    coll = size > a * (1 - eccs)
    mask = _transits(Ptra, incs, parallel) & (coll == False)  # noqa: E712
    if extra is not None:
        mask = mask & extra
    a_col = a if isinstance(a, np.ndarray) else np.full(N, a)
    cols = (rps, P_orb, incs, a_col, R_host, u1, u2, eccs, argps,
            0.0 if fr_comp is None else fr_comp)
    best, lnZ = _evidence(MODEL_TP, is_host, parallel, time, flux, sigma, cols, mask, lnprior, N,
                         exptime, nsamples)
    return _table(best, lnZ, M_s=M_host, R_s=R_host, u1=u1, u2=u2, P_orb=P_orb, inc=incs, b=b,
                  R_p=rps, ecc=eccs, argp=argps, M_EB=None, R_EB=None, fluxratio_EB=None,
                  fluxratio_comp=fr_comp)


def _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs, eccs,
                     argps, masses, radii, fluxratios, M_host, R_host, u1, u2, fr_comp, is_host,
                     extra, lnprior, twin_is_host_copy=False):
    """common tail of the *EB scenarios: q < 0.95 at P_orb and q >= 0.95 at 2 P_orb.
    twin_is_host_copy: lnZ_NEB_evolved treats the twin as a copy of the host (2 R_host in its
    transit probability, R_host as its radius; marginal_likelihoods.py:3052, 3100)."""
    e_corr = _e_corr(eccs, argps)
    a = _sma(M_host + masses, P_orb)
    size = radii * Rsun + R_host * Rsun
    Ptra = size / a * e_corr
    a_twin = _sma(M_host + masses, 2 * P_orb)
    size_twin = (R_host * Rsun + R_host * Rsun) if twin_is_host_copy else size
    Ptra_twin = size_twin / a_twin * e_corr
    b = _impact(a, eccs, argps, incs, R_host)
    b_twin = _impact(a_twin, eccs, argps, incs, R_host)
    coll = size > a * (1 - eccs)
    coll_twin = (2 * R_host * Rsun) > a_twin * (1 - eccs)

    hit = _transits(Ptra, incs, parallel)
    hit_twin = _transits(Ptra_twin, incs, parallel)
    if not parallel:
        hit_twin = hit_twin & (Ptra <= 1)   # the serial loop `continue`s before the twin test
    mask = hit & (coll == False) & (qs < 0.95)  # noqa: E712
    mask_twin = hit_twin & (coll_twin == False) & (qs >= 0.95)  # noqa: E712
    if extra is not None:
        mask, mask_twin = mask & extra, mask_twin & extra
    frc = 0.0 if fr_comp is None else fr_comp
    out = []
    r_twin = R_host if twin_is_host_copy else radii
    for model, m, per, sma, bb, r_eb in ((MODEL_EB, mask, P_orb, a, b, radii),
                                         (MODEL_EB_TWIN, mask_twin, 2 * P_orb, a_twin, b_twin, r_twin)):
        cols = (r_eb, fluxratios, per, incs, sma, R_host, u1, u2, eccs, argps, frc)
        best, lnZ = _evidence(model, is_host, parallel, time, flux, sigma, cols, m, lnprior, N,
                             exptime, nsamples)
        out.append(_table(best, lnZ, M_s=M_host, R_s=R_host, u1=u1, u2=u2, P_orb=per, inc=incs,
                          b=bb, R_p=None, ecc=eccs, argp=argps, M_EB=masses, R_EB=r_eb,
                          fluxratio_EB=fluxratios, fluxratio_comp=fr_comp))
    return out[0], out[1]


def _draw_planet(N, M_for_rp, P_orb, flatpriors):
    rps = sample_rp(np.random.rand(N), M_for_rp, flatpriors)
    incs = sample_inc(np.random.rand(N))
    eccs = sample_ecc(np.random.rand(N), planet=True, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    return rps, incs, eccs, argps


# ---------------------------------------------------------------------------------------
# target-star scenarios
def lnZ_TTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N: int = 1000000, parallel: bool = False,
            mission: str = "TESS", flatpriors: bool = False, exptime: float = 0.00139,
            nsamples: int = 20):
    """Transiting planet on the target star (also the NTP scenario of a nearby star)."""
    P_orb = _periods(P_orb, N)
    a = _sma(M_s, P_orb)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    rps, incs, eccs, argps = _draw_planet(N, np.full(N, M_s), P_orb, flatpriors)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_s, R_s, u1, u2, None, False, None, None)


def lnZ_TEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N: int = 1000000, parallel: bool = False,
            mission: str = "TESS", flatpriors: bool = False, exptime: float = 0.00139,
            nsamples: int = 20):
    """Eclipsing binary on the target star (also NEB / NEBx2P of a nearby star)."""
    P_orb = _periods(P_orb, N)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), M_s)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    masses = qs * M_s
    radii, _ = stellar_relations(masses, np.full(N, R_s), np.full(N, Teff))
    fluxratios = _flux_share(masses, M_s)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_s, R_s, u1, u2, None, False,
                            None, None)


def lnZ_PTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file: str = None,
            filt: str = "TESS", N: int = 1000000, parallel: bool = False, mission: str = "TESS",
            flatpriors: bool = False, exptime: float = 0.00139, nsamples: int = 20,
            molusc_file: str = None):
    """Planet on the target, diluted by an unresolved bound companion."""
    P_orb = _periods(P_orb, N)
    a = _sma(M_s, P_orb)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    qs_comp = _bound_companions(M_s, N, molusc_file)
    masses_comp = qs_comp * M_s
    fr_comp = _flux_share(masses_comp, M_s)
    lnprior = _bound_prior("TP", M_s, plx, N, molusc_file, contrast_curve_file,
                           fr_comp / (1 - fr_comp),
                           lambda: (lambda f: f / (1 - f))(_flux_share(masses_comp, M_s, filt)))
    rps, incs, eccs, argps = _draw_planet(N, np.full(N, M_s), P_orb, flatpriors)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_s, R_s, u1, u2, fr_comp, False, qs_comp != 0.0, lnprior)


def lnZ_PEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file: str = None,
            filt: str = "TESS", N: int = 1000000, parallel: bool = False, mission: str = "TESS",
            flatpriors: bool = False, exptime: float = 0.00139, nsamples: int = 20,
            molusc_file: str = None):
    """EB on the target, diluted by an unresolved bound companion."""
    P_orb = _periods(P_orb, N)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), M_s)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    qs_comp = _bound_companions(M_s, N, molusc_file)
    masses = qs * M_s
    radii, _ = stellar_relations(masses, np.full(N, R_s), np.full(N, Teff))
    fluxratios = _flux_share(masses, M_s)
    masses_comp = qs_comp * M_s
    fr_comp = _flux_share(masses_comp, M_s)
    lnprior = _bound_prior("EB", M_s, plx, N, molusc_file, contrast_curve_file,
                           fr_comp / (1 - fr_comp),
                           lambda: (lambda f: f / (1 - f))(_flux_share(masses_comp, M_s, filt)))
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_s, R_s, u1, u2, fr_comp,
                            False, qs_comp != 0.0, lnprior)


def _companion_host(M_s, R_s, Teff, Z, N, mission, molusc_file, teff_cap):
    """properties of an unresolved bound companion acting as the host (S scenarios)"""
    qs_comp = _bound_companions(M_s, N, molusc_file)
    masses_comp = qs_comp * M_s
    radii_comp, Teffs_comp = stellar_relations(masses_comp, np.full(N, R_s), np.full(N, Teff))
    loggs_comp = _logg(masses_comp, radii_comp)
    fr_comp = _flux_share(masses_comp, M_s)
    u1s, u2s = _ldc(mission).companions(Z, Teffs_comp, loggs_comp, teff_cap)
    return qs_comp, masses_comp, radii_comp, Teffs_comp, fr_comp, u1s, u2s


def lnZ_STP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file: str = None,
            filt: str = "TESS", N: int = 1000000, parallel: bool = False, mission: str = "TESS",
            flatpriors: bool = False, exptime: float = 0.00139, nsamples: int = 20,
            molusc_file: str = None):
    """Planet on an unresolved bound companion of the target."""
    P_orb = _periods(P_orb, N)
    qs_comp, masses_comp, radii_comp, _, fr_comp, u1s, u2s = _companion_host(
        M_s, R_s, Teff, Z, N, mission, molusc_file, 10000)
    lnprior = _bound_prior("TP", M_s, plx, N, molusc_file, contrast_curve_file,
                           fr_comp / (1 - fr_comp),
                           lambda: (lambda f: f / (1 - f))(_flux_share(masses_comp, M_s, filt)))
    rps, incs, eccs, argps = _draw_planet(N, masses_comp, P_orb, flatpriors)
    a = _sma(masses_comp, P_orb)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, masses_comp, radii_comp, u1s, u2s, fr_comp, True,
                          qs_comp != 0.0, lnprior)


def lnZ_SEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file: str = None,
            filt: str = "TESS", N: int = 1000000, parallel: bool = False, mission: str = "TESS",
            flatpriors: bool = False, exptime: float = 0.00139, nsamples: int = 20,
            molusc_file: str = None):
    """EB on an unresolved bound companion of the target."""
    P_orb = _periods(P_orb, N)
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), M_s)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    qs_comp, masses_comp, radii_comp, Teffs_comp, fr_comp, u1s, u2s = _companion_host(
        M_s, R_s, Teff, Z, N, mission, molusc_file, 13000)
    masses = qs * masses_comp
    radii, _ = stellar_relations(masses, radii_comp, Teffs_comp)
    fluxratios = _flux_share(masses, M_s)

    def cc_term():
        f_eb, f_c = _flux_share(masses, M_s, filt), _flux_share(masses_comp, M_s, filt)
        return (f_c / (1 - f_c)) + (f_eb / (1 - f_eb))

    lnprior = _bound_prior("EB", M_s, plx, N, molusc_file, contrast_curve_file,
                           (fr_comp / (1 - fr_comp)) + (fluxratios / (1 - fluxratios)), cc_term)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, masses_comp, radii_comp, u1s,
                            u2s, fr_comp, True, qs_comp != 0.0, lnprior)


# ---------------------------------------------------------------------------------------
# chance-aligned field stars (TRILEGAL population)
def lnZ_DTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file: str = None, filt: str = "TESS", N: int = 1000000,
            parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
            exptime: float = 0.00139, nsamples: int = 20):
    """Planet on the target, diluted by an unresolved background star."""
    P_orb = _periods(P_orb, N)
    a = _sma(M_s, P_orb)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    field = _Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
    idxs = np.random.randint(0, field.N_comp - 1, N)   # sic: the last star is never drawn
    lnprior = field.prior(N, idxs, contrast_curve_file, filt)
    rps, incs, eccs, argps = _draw_planet(N, np.full(N, M_s), P_orb, flatpriors)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_s, R_s, u1, u2, field.fluxratios[idxs], False, None,
                          lnprior)


def lnZ_DEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file: str = None, filt: str = "TESS", N: int = 1000000,
            parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
            exptime: float = 0.00139, nsamples: int = 20):
    """EB on the target, diluted by an unresolved background star."""
    P_orb = _periods(P_orb, N)
    u1, u2 = _ldc(mission).star(Z, Teff, _logg(M_s, R_s))
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), M_s)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    masses = qs * M_s
    radii, _ = stellar_relations(masses, np.full(N, R_s), np.full(N, Teff))
    fluxratios = _flux_share(masses, M_s)
    field = _Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
    idxs = np.random.randint(0, field.N_comp - 1, N)
    lnprior = field.prior(N, idxs, contrast_curve_file, filt)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_s, R_s, u1, u2,
                            field.fluxratios[idxs], False, None, lnprior)


def lnZ_BTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file: str = None, filt: str = "TESS", N: int = 1000000,
            parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
            exptime: float = 0.00139, nsamples: int = 20):
    """Planet on an unresolved background star."""
    P_orb = _periods(P_orb, N)
    field = _Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
    radii_f = field.radii()
    u1f, u2f = _ldc(mission).field_stars(field.Teffs, field.loggs, field.Zs)
    idxs = np.random.randint(0, field.N_comp, N)
    lnprior = field.prior(N, idxs, contrast_curve_file, filt)
    M_host, R_host = field.masses[idxs], radii_f[idxs]
    rps, incs, eccs, argps = _draw_planet(N, M_host, P_orb, flatpriors)
    a = _sma(M_host, P_orb)
    extra = (field.loggs[idxs] >= 3.5) & (field.Teffs[idxs] <= 10000)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_host, R_host, u1f[idxs], u2f[idxs],
                          field.fluxratios[idxs], True, extra, lnprior)


def lnZ_BEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file: str = None, filt: str = "TESS", N: int = 1000000,
            parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
            exptime: float = 0.00139, nsamples: int = 20):
    """EB on an unresolved background star."""
    P_orb = _periods(P_orb, N)
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), M_s)
    sample_q_companion(np.random.rand(N), M_s)   # drawn and unused in the reference (:2089)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    field = _Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
    radii_f = field.radii()
    u1f, u2f = _ldc(mission).field_stars(field.Teffs, field.loggs, field.Zs)
    idxs = np.random.randint(0, field.N_comp, N)
    M_host, R_host = field.masses[idxs], radii_f[idxs]
    masses = qs * M_host
    radii, _ = stellar_relations(masses, R_host, field.Teffs[idxs])
    fr_comp = field.fluxratios[idxs]
    # the background star sits at another distance: rescale the bound-pair flux share
    fluxratios = _flux_share(masses, M_s) * (fr_comp / _flux_share(M_host, M_s))
    fr_term = (fr_comp / (1 - fr_comp)) + (fluxratios / (1 - fluxratios))
    fr_term_cc = None
    if contrast_curve_file is not None:
        fr_comp_cc = field.band_fluxratio(filt)[idxs]
        fluxratios_cc = _flux_share(masses, M_s, filt) * (fr_comp_cc / _flux_share(M_host, M_s, filt))
        fr_term_cc = (fr_comp_cc / (1 - fr_comp_cc)) + (fluxratios_cc / (1 - fluxratios_cc))
    lnprior = field.prior(N, idxs, contrast_curve_file, filt, fr_term, fr_term_cc)
    extra = (field.loggs[idxs] >= 3.5) & (field.Teffs[idxs] <= 10000)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_host, R_host, u1f[idxs],
                            u2f[idxs], fr_comp, True, extra, lnprior)


# ---------------------------------------------------------------------------------------
# exported by the reference but never called by calc_probs
def _similar_field_stars(trilegal_fname, Tmag, mission):
    """TRILEGAL stars within one magnitude of Tmag: the possible identities of a nearby star
    of unknown properties (marginal_likelihoods.py:2402-2446)"""
    Tmags, masses, loggs, Teffs, Zs, _, _, _ = trilegal_results(trilegal_fname, Tmag)
    near = (Tmag - 1 < Tmags) & (Tmags < Tmag + 1)
    masses, loggs, Teffs, Zs = masses[near], loggs[near], Teffs[near], Zs[near]
    radii = np.sqrt(G * masses * Msun / 10 ** loggs) / Rsun
    u1s, u2s = (_ldc(mission).field_stars(Teffs, loggs, Zs) if masses.size
                else (np.zeros(0), np.zeros(0)))
    return masses, radii, loggs, Teffs, u1s, u2s


_EMPTY_KEYS = ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "R_p", "ecc", "argp", "M_EB", "R_EB",
               "fluxratio_EB", "fluxratio_comp")


def _empty_result(with_b):
    """what the reference returns when no similar field star exists: scalar zeros and
    lnZ = -inf; lnZ_NTP_unknown's dict has no 'b' entry, lnZ_NEB_unknown's has
    (marginal_likelihoods.py:2450-2468, 2645-2665)"""
    res = {k: 0 for k in _EMPTY_KEYS}
    if with_b:
        res["b"] = 0
    res["lnZ"] = -np.inf
    return res


def lnZ_NTP_unknown(time, flux, sigma, P_orb, Tmag, trilegal_fname, N: int = 1000000,
                    parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
                    exptime: float = 0.00139, nsamples: int = 20):
    """Planet on a nearby star of unknown properties (marginal_likelihoods.py:2365-2551)."""
    P_orb = _periods(P_orb, N)
    masses_f, radii_f, loggs_f, Teffs_f, u1f, u2f = _similar_field_stars(trilegal_fname, Tmag, mission)
    if masses_f.size == 0:
        return _empty_result(with_b=False)
    idxs = np.random.randint(0, masses_f.size, N)
    M_host, R_host = masses_f[idxs], radii_f[idxs]
    rps, incs, eccs, argps = _draw_planet(N, M_host, P_orb, flatpriors)
    a = _sma(M_host, P_orb)
    extra = (loggs_f[idxs] >= 3.5) & (Teffs_f[idxs] <= 10000)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_host, R_host, u1f[idxs], u2f[idxs], None, False, extra,
                          None)


def lnZ_NEB_unknown(time, flux, sigma, P_orb, Tmag, trilegal_fname, N: int = 1000000,
                    parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
                    exptime: float = 0.00139, nsamples: int = 20):
    """EB on a nearby star of unknown properties (marginal_likelihoods.py:2554-2829).  Like the
    reference, an empty field population returns ONE dict although the normal return is two."""
    P_orb = _periods(P_orb, N)
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), 1.0)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    masses_f, radii_f, loggs_f, Teffs_f, u1f, u2f = _similar_field_stars(trilegal_fname, Tmag, mission)
    if masses_f.size == 0:
        return _empty_result(with_b=True)
    idxs = np.random.randint(0, masses_f.size, N)
    M_host, R_host = masses_f[idxs], radii_f[idxs]
    masses = qs * M_host
    radii, _ = stellar_relations(masses, R_host, Teffs_f[idxs])
    f_eb = flux_relation(masses)
    fluxratios = f_eb / (f_eb + flux_relation(M_host))
    extra = (loggs_f[idxs] >= 3.5) & (Teffs_f[idxs] <= 10000)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_host, R_host, u1f[idxs],
                            u2f[idxs], None, False, extra, None)


def _evolved_host(R_s, Teff, Z, mission):
    """a subgiant of log g = 3.0 and radius R_s (marginal_likelihoods.py:2868-2891)"""
    logg = 3.0
    M_s = (10 ** logg) * (R_s * Rsun) ** 2 / G / Msun
    u1, u2 = _ldc(mission).star(Z, Teff, logg)
    return M_s, u1, u2


def lnZ_NTP_evolved(time, flux, sigma, P_orb, R_s, Teff, Z, N: int = 1000000,
                    parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
                    exptime: float = 0.00139, nsamples: int = 20):
    """Planet on an evolved nearby star (marginal_likelihoods.py:2832-2966)."""
    P_orb = _periods(P_orb, N)
    M_s, u1, u2 = _evolved_host(R_s, Teff, Z, mission)
    rps, incs, eccs, argps = _draw_planet(N, np.full(N, M_s), P_orb, flatpriors)
    a = _sma(M_s, P_orb)
    return _planet_branch(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, rps, incs,
                          eccs, argps, a, M_s, R_s, u1, u2, None, False, None, None)


def lnZ_NEB_evolved(time, flux, sigma, P_orb, R_s, Teff, Z, N: int = 1000000,
                    parallel: bool = False, mission: str = "TESS", flatpriors: bool = False,
                    exptime: float = 0.00139, nsamples: int = 20):
    """EB on an evolved nearby star (marginal_likelihoods.py:2969-3178).  Reference quirk kept:
    the twin is a copy of the host (2 R_s in Ptra_twin, R_s as its radius)."""
    P_orb = _periods(P_orb, N)
    M_s, u1, u2 = _evolved_host(R_s, Teff, Z, mission)
    incs = sample_inc(np.random.rand(N))
    qs = sample_q(np.random.rand(N), 1.0)
    eccs = sample_ecc(np.random.rand(N), planet=False, P_orb=np.mean(P_orb))
    argps = sample_w(np.random.rand(N))
    masses = qs * M_s
    radii, _ = stellar_relations(masses, np.full(N, R_s), np.full(N, Teff))
    fluxratios = _flux_share(masses, M_s)
    return _binary_branches(time, flux, sigma, N, parallel, exptime, nsamples, P_orb, qs, incs,
                            eccs, argps, masses, radii, fluxratios, M_s, R_s, u1, u2, None, False,
                            None, None, twin_is_host_copy=True)


# ---------------------------------------------------------------------------------------
# where the priors are sampled:
#   "device"        (default) the whole scenario on the GPU: one HIP kernel draws (Philox4x32-10, keyed by
#                   torch's CPU generator: torch.manual_seed reproduces a run), derives, masks and weighs the
#                   N draws; statistically equivalent to the reference (tests/test_gpu_equivalence.py,
#                   tests/test_gpu_notebook_anchors.py), ~300x faster end to end than "numpy" at N = 1e6
#   "numpy"         host numpy, the reference's own arithmetic on numpy's global stream: draw-for-draw and
#                   bit-for-bit the reference under the same np.random.seed -- the validation mode
#   "numpy-device"  numpy's global stream supplies the uniforms in the reference's order, everything
#                   downstream runs on the GPU: the same draws as the reference, derived columns equal to
#                   rounding, ~6x faster than "numpy"
# TRX_SAMPLING in the environment overrides the default.
import os as _os

DEFAULT_SAMPLING = "device"
_sampling = {"mode": _os.environ.get("TRX_SAMPLING", DEFAULT_SAMPLING)}
if _sampling["mode"] not in ("numpy", "numpy-device", "device"):
    raise ValueError("TRX_SAMPLING must be 'numpy', 'numpy-device' or 'device'")
if _sampling["mode"] == "numpy-device":
    from . import device_pipeline as _dp
    _dp.RNG = _dp.NumpyStreamRng()


def set_sampling(mode):
    if mode not in ("numpy", "numpy-device", "device"):
        raise ValueError("sampling mode must be 'numpy', 'numpy-device' or 'device'")
    if mode != "numpy":
        from . import device_pipeline
        device_pipeline.RNG = (device_pipeline.NumpyStreamRng() if mode == "numpy-device"
                               else device_pipeline.TorchRng())
    _sampling["mode"] = mode


def _dispatch(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        mode = _sampling["mode"]
        if mode != "numpy":
            # one fused HIP kernel per scenario (fused.py)
            from . import fused
            return getattr(fused, fn.__name__)(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


for _name in ("lnZ_TTP", "lnZ_TEB", "lnZ_PTP", "lnZ_PEB", "lnZ_STP", "lnZ_SEB", "lnZ_DTP", "lnZ_DEB",
              "lnZ_BTP", "lnZ_BEB"):
    globals()[_name] = _dispatch(globals()[_name])
