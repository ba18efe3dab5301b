"""ctypes/numpy front-end of the CPU oracle (oracle/trx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under triceratops_amd/ may import this.

`QuadraticModel` below has the three methods the reference calls on
pytransit.QuadraticModel (likelihoods.py:24-25, 61-71, 348-349, 414-422), so the
*unmodified* reference can be run on top of the oracle inside the build container
(tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtrx_oracle.so")


def use_library(path):
    """load another build of trx_oracle.c from now on (bench.py's cpu_baseline compiles one with
    -O3 -march=native on the box it runs on)"""
    global _SO, _lib
    _SO, _lib = path, None

MODEL_TP, MODEL_EB, MODEL_EB_TWIN = 0, 1, 2
FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K = 1, 2
N_PARAM = {MODEL_TP: 10, MODEL_EB: 11, MODEL_EB_TWIN: 11}

_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    src = os.path.join(_HERE, "trx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libtrx_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.trxo_ma_flux.restype = ctypes.c_double
        L.trxo_ma_flux.argtypes = [ctypes.c_double] * 4
        L.trxo_kepler_E.restype = ctypes.c_double
        L.trxo_kepler_E.argtypes = [ctypes.c_double] * 2
        L.trxo_evaluate_pv.restype = None
        L.trxo_evaluate_pv.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_long,
                                       ctypes.c_double, ctypes.c_int, _dp]
        L.trxo_lnl_batch.restype = None
        L.trxo_lnl_batch.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                     ctypes.c_double, _dp, ctypes.c_long, ctypes.c_double,
                                     ctypes.c_int, _dp]
        L.trxo_flux_grid.restype = None
        L.trxo_flux_grid.argtypes = [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp,
                                     ctypes.c_long, ctypes.c_double, ctypes.c_int, _dp, _dp]
        L.trxo_chi2_grid.restype = None
        L.trxo_chi2_grid.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_long, ctypes.c_double, _dp]
        L.trxo_log_mean_exp.restype = ctypes.c_int
        L.trxo_log_mean_exp.argtypes = [_dp, ctypes.c_long, ctypes.c_long, _dp]
        L.trxo_normalize_probabilities.restype = ctypes.c_int
        L.trxo_normalize_probabilities.argtypes = [_dp, ctypes.c_int, _dp]
        L.trxo_num_threads.restype = ctypes.c_int
        L.trxo_set_num_threads.argtypes = [ctypes.c_int]
        if hasattr(L, "trxo_set_window_skip"):      # only in bench.py's -DTRXO_WINDOW_SKIP build
            L.trxo_set_window_skip.argtypes = [ctypes.c_int]
            L.trxo_set_window_skip.restype = None
        _lib = L
    return _lib


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def ma_flux(z, k, u1, u2):
    f = np.vectorize(lambda zz, kk, a, b: lib().trxo_ma_flux(zz, kk, a, b), otypes=[float])
    return f(z, k, u1, u2)


def kepler_E(M, e):
    f = np.vectorize(lambda m, ee: lib().trxo_kepler_E(m, ee), otypes=[float])
    return f(M, e)


def evaluate_pv(time, pvp, ldc, exptime=0.0, nsamples=1):
    time, tp = _c(time)
    pvp, pp = _c(np.atleast_2d(pvp))
    ldc, lp = _c(np.atleast_2d(ldc))
    n = pvp.shape[0]
    assert pvp.shape[1] == 7 and ldc.shape == (n, 2)
    out = np.empty((n, time.size))
    lib().trxo_evaluate_pv(tp, time.size, pp, lp, n, float(exptime), int(nsamples),
                           out.ctypes.data_as(_dp))
    return out


def pack_params(model, *cols):
    """SoA [n_param][n] block from per-sample columns (reference argument order)."""
    assert len(cols) == N_PARAM[model]
    n = max(np.size(c) for c in cols)
    return np.ascontiguousarray(
        np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), (n,)) for c in cols]))


def lnl_batch(model, time, flux, sigma, params, companion_is_host=False, exptime=0.00139,
              nsamples=20, scalar_k=False):
    time, tp = _c(time)
    flux, fp = _c(flux)
    params, pp = _c(params)
    assert params.shape[0] == N_PARAM[model]
    n = params.shape[1]
    out = np.empty(n)
    flags = (FLAG_COMPANION_IS_HOST if companion_is_host else 0) | (FLAG_SCALAR_K if scalar_k else 0)
    lib().trxo_lnl_batch(model, flags, tp, fp, time.size, float(sigma), pp, n, float(exptime),
                         int(nsamples), out.ctypes.data_as(_dp))
    return out


def flux_grid(model, time, params, companion_is_host=False, exptime=0.00139, nsamples=20,
              scalar_k=False):
    time, tp = _c(time)
    params, pp = _c(params)
    n = params.shape[1]
    out = np.empty((n, time.size))
    sd = np.zeros(n)
    flags = (FLAG_COMPANION_IS_HOST if companion_is_host else 0) | (FLAG_SCALAR_K if scalar_k else 0)
    lib().trxo_flux_grid(model, flags, tp, time.size, pp, n, float(exptime), int(nsamples),
                         out.ctypes.data_as(_dp), sd.ctypes.data_as(_dp))
    return out, sd


def chi2_grid(flux, model_grid, sigma):
    flux, fp = _c(flux)
    model_grid, mp_ = _c(model_grid)
    n, nt = model_grid.shape
    out = np.empty(n)
    lib().trxo_chi2_grid(fp, mp_, nt, n, float(sigma), out.ctypes.data_as(_dp))
    return out


def log_mean_exp(logw, N_total):
    logw, lp = _c(np.ravel(logw))
    out = ctypes.c_double()
    st = lib().trxo_log_mean_exp(lp, logw.size, int(N_total), ctypes.byref(out))
    if st:
        raise ValueError("N_total (%d) must equal len(logw) (%d)" % (N_total, logw.size))
    return out.value


def normalize_probabilities(lnZ):
    lnZ, lp = _c(np.ravel(lnZ))
    probs = np.zeros(lnZ.size)
    st = lib().trxo_normalize_probabilities(lp, lnZ.size, probs.ctypes.data_as(_dp))
    return probs, ("ok", "all_neginf", "anomaly")[st]


def num_threads():
    return lib().trxo_num_threads()


def set_window_skip(on):
    """bench.py's cpu_baseline only: exposures outside the transit window return 1 without an orbit solve (the
    GPU kernels' early-out); the plain restatement (default) evaluates every point like the reference"""
    lib().trxo_set_window_skip(int(bool(on)))


def set_num_threads(n):
    lib().trxo_set_num_threads(int(n))


class QuadraticModel:
    """Oracle stand-in for pytransit.QuadraticModel at the three call shapes the
    reference uses (likelihoods.py:24-25, 61-71, 348-349, 414-422)."""

    def __init__(self, interpolate=False, **_):
        self.time = None
        self.exptime = 0.0
        self.nsamples = 1

    def set_data(self, time, lcids=None, pbids=None, nsamples=None, exptimes=None, **_):
        self.time = np.ascontiguousarray(time, dtype=np.float64)
        self.nsamples = int(np.ravel(nsamples)[0]) if nsamples is not None else 1
        self.exptime = float(np.ravel(exptimes)[0]) if exptimes is not None else 0.0

    def evaluate_pv(self, pvp, ldc):
        return evaluate_pv(self.time, pvp, ldc, self.exptime, self.nsamples)

    def evaluate_ps(self, k, ldc, t0, p, a, i, e=0.0, w=0.0):
        pvp = np.array([[k, t0, p, a, i, e, w]], dtype=np.float64)
        return evaluate_pv(self.time, pvp, np.asarray(ldc, dtype=np.float64).reshape(1, 2),
                           self.exptime, self.nsamples)[0]
