"""Light-curve simulators and chi^2 likelihoods on the MI355X kernels.

Same names, argument meaning and return values as the reference's triceratops/likelihoods.py:
  simulate_TP_transit (27-80), simulate_EB_transit (83-160), lnL_TP (164-204), lnL_EB (207-253),
  lnL_EB_twin (256-299)                                   -- scalar path (`abs(k-1)<1e-6` rule)
  simulate_TP_transit_p (302-358), simulate_EB_transit_p (361-439), lnL_TP_p (443-487),
  lnL_EB_p (490-539), lnL_EB_twin_p (542-587)             -- vector path
Every function packs its per-sample arguments into one SoA parameter block, runs libtrx on the
current CUDA device and returns numpy arrays.  lnL_* return +chi^2/2 (a negative log-likelihood
without the Gaussian constant); +inf marks an EB draw whose secondary eclipse is deeper than
1.5 sigma.  Differences from the reference, on purpose: the caller's `inc` array is not
converted to radians in place (likelihoods.py:344, 410), and the functions do not go through
shared mutable model objects, so they are re-entrant.  The module-level names `tm` and `tm_sec`
(likelihoods.py:24-25) exist for code that imports them, but nothing here calls `set_data` on
them.
"""
import numpy as np

from . import _lib
from ._lib import FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K, MODEL_EB, MODEL_EB_TWIN, MODEL_TP
from .constants import G, Msun, Rearth, Rsun, au, pi  # noqa: F401  (reference module exports)
from .transit_model import QuadraticModel

tm = QuadraticModel(interpolate=False)
tm_sec = QuadraticModel(interpolate=False)


def _flags(companion_is_host, scalar):
    return (FLAG_COMPANION_IS_HOST if companion_is_host else 0) | (FLAG_SCALAR_K if scalar else 0)


def _block(model, cols):
    n = max(int(np.size(c)) for c in cols)
    return _lib.dev(_lib.pack_params(model, cols, n))


def _grid(model, time, cols, companion_is_host, exptime, nsamples, scalar):
    grid, sec = _lib.flux_grid(model, _flags(companion_is_host, scalar), _lib.dev(time),
                               _block(model, cols), exptime, nsamples, want_secdepth=True)
    return grid.cpu().numpy(), sec.cpu().numpy()


def _halfchi2(model, time, flux, sigma, cols, companion_is_host, exptime, nsamples, scalar):
    out = _lib.lnl_batch(model, _flags(companion_is_host, scalar), _lib.dev(time), _lib.dev(flux),
                         sigma, _block(model, cols), exptime, nsamples)
    return out.cpu().numpy()


# ---------------------------------------------------------------------------------------
# vector path
def simulate_TP_transit_p(time, R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                          companion_fluxratio, companion_is_host: bool = False,
                          exptime: float = 0.00139, nsamples: int = 20):
    """Diluted transiting-planet light curves, one row per sample: (n, n_time)."""
    cols = (R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return _grid(MODEL_TP, time, cols, companion_is_host, exptime, nsamples, False)[0]


def simulate_EB_transit_p(time, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                          companion_fluxratio, companion_is_host: bool = False,
                          exptime: float = 0.00139, nsamples: int = 20):
    """Diluted eclipsing-binary light curves (n, n_time) and secondary depths (n, 1)."""
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    grid, sec = _grid(MODEL_EB, time, cols, companion_is_host, exptime, nsamples, False)
    return grid, sec.reshape(-1, 1)


def lnL_TP_p(time, flux, sigma, R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
             companion_fluxratio, companion_is_host: bool = False,
             exptime: float = 0.00139, nsamples: int = 20):
    """chi^2/2 of each transiting-planet sample against (time, flux, sigma): (n,)."""
    cols = (R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return _halfchi2(MODEL_TP, time, flux, sigma, cols, companion_is_host, exptime, nsamples, False)


def lnL_EB_p(time, flux, sigma, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
             companion_fluxratio, companion_is_host: bool = False,
             exptime: float = 0.00139, nsamples: int = 20):
    """chi^2/2 of each EB sample (q < 0.95); +inf where the secondary depth is >= 1.5 sigma."""
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return _halfchi2(MODEL_EB, time, flux, sigma, cols, companion_is_host, exptime, nsamples, False)


def lnL_EB_twin_p(time, flux, sigma, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                  companion_fluxratio, companion_is_host: bool = False,
                  exptime: float = 0.00139, nsamples: int = 20):
    """chi^2/2 of each twin-EB sample (q >= 0.95, called with 2 x P_orb); no secondary cut."""
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return _halfchi2(MODEL_EB_TWIN, time, flux, sigma, cols, companion_is_host, exptime, nsamples,
                     False)


# ---------------------------------------------------------------------------------------
# scalar path (one sample; the reference applies `abs(k - 1) < 1e-6` and 1/k here)
def simulate_TP_transit(time, R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                        companion_fluxratio: float = 0.0, companion_is_host: bool = False,
                        exptime: float = 0.00139, nsamples: int = 20):
    cols = (R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return _grid(MODEL_TP, time, cols, companion_is_host, exptime, nsamples, True)[0][0]


def simulate_EB_transit(time, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                        companion_fluxratio: float = 0.0, companion_is_host: bool = False,
                        exptime: float = 0.00139, nsamples: int = 20):
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    grid, sec = _grid(MODEL_EB, time, cols, companion_is_host, exptime, nsamples, True)
    return grid[0], float(sec[0])


def lnL_TP(time, flux, sigma, R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
           companion_fluxratio: float = 0.0, companion_is_host: bool = False,
           exptime: float = 0.00139, nsamples: int = 20):
    cols = (R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return float(_halfchi2(MODEL_TP, time, flux, sigma, cols, companion_is_host, exptime,
                           nsamples, True)[0])


def lnL_EB(time, flux, sigma, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
           companion_fluxratio: float = 0.0, companion_is_host: bool = False,
           exptime: float = 0.00139, nsamples: int = 20):
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return float(_halfchi2(MODEL_EB, time, flux, sigma, cols, companion_is_host, exptime,
                           nsamples, True)[0])


def lnL_EB_twin(time, flux, sigma, R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                companion_fluxratio: float = 0.0, companion_is_host: bool = False,
                exptime: float = 0.00139, nsamples: int = 20):
    cols = (R_EB, EB_fluxratio, P_orb, inc, a, R_s, u1, u2, ecc, argp, companion_fluxratio)
    return float(_halfchi2(MODEL_EB_TWIN, time, flux, sigma, cols, companion_is_host, exptime,
                           nsamples, True)[0])
