"""The three host-pointer entry points of include/trx.h, called exactly as INTEGRATION.md
section C binds them (ctypes on numpy arrays, no torch anywhere in the call), against the CPU
oracle.  These are the functions a reference maintainer's `triceratops/_trx.py` would call."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")   # only so that libtrx binds to the HIP runtime torch loaded

from oracle import oracle as O
from triceratops_amd import _lib, synth

pytestmark = pytest.mark.gpu

# ---- the binding of INTEGRATION.md section C, verbatim ------------------------------------------
_L = ctypes.CDLL(_lib.LIB_PATH)
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_L.trx_lnl_batch_host.restype = ctypes.c_int
_L.trx_lnl_batch_host.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                  ctypes.c_double, _dp, ctypes.c_long, ctypes.c_double,
                                  ctypes.c_int, _dp]
_L.trx_flux_grid_host.restype = ctypes.c_int
_L.trx_flux_grid_host.argtypes = [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp,
                                  ctypes.c_long, ctypes.c_double, ctypes.c_int, _dp, _dp]
_L.trx_log_mean_exp_host.restype = ctypes.c_int
_L.trx_log_mean_exp_host.argtypes = [_dp, ctypes.c_long, ctypes.c_long, _dp]
_L.trx_last_error.restype = ctypes.c_char_p
TP, EB, EB_TWIN = 0, 1, 2            # TRX_MODEL_*
IS_HOST, SCALAR_K = 1, 2             # TRX_FLAG_*


def lnl(model, time, flux, sigma, cols, companion_is_host=False, exptime=0.00139, nsamples=20):
    n = max(np.size(c) for c in cols)
    block = np.ascontiguousarray(np.stack([np.broadcast_to(c, (n,)) for c in cols]), dtype=np.float64)
    out = np.empty(n)
    rc = _L.trx_lnl_batch_host(model, IS_HOST if companion_is_host else 0,
                               np.ascontiguousarray(time, dtype=np.float64),
                               np.ascontiguousarray(flux, dtype=np.float64), len(time),
                               float(sigma), block, n, float(exptime), int(nsamples), out)
    if rc:
        raise RuntimeError(_L.trx_last_error().decode())
    return out


def lnL_TP_p(time, flux, sigma, R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
             companion_fluxratio, companion_is_host=False, exptime=0.00139, nsamples=20):
    return lnl(TP, time, flux, sigma, (R_p, P_orb, inc, a, R_s, u1, u2, ecc, argp,
                                       companion_fluxratio), companion_is_host, exptime, nsamples)


# --------------------------------------------------------------------------------------------
def _lc(n_time):
    rng = np.random.default_rng(synth.SEED + 77)
    t = synth.time_grid(n_time)
    curve = O.flux_grid(O.MODEL_TP, t, synth.reference_tp_row())[0][0]
    return rng, t, synth.noisy_light_curve(rng, curve)


def _same(got, want, rtol):
    assert np.array_equal(np.isposinf(want), np.isposinf(got))
    fin = np.isfinite(want)
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < rtol


@pytest.mark.parametrize("is_host", [False, True])
def test_lnl_batch_host_as_the_reference_would_call_it(is_host):
    rng, t, flux = _lc(120)
    rows = synth.tp_rows(rng, 300, has_companion=True)
    got = lnL_TP_p(t, flux, synth.SIGMA, *rows, companion_is_host=is_host)
    want = O.lnl_batch(O.MODEL_TP, t, flux, synth.SIGMA, rows, companion_is_host=is_host)
    _same(got, want, 1e-9)


@pytest.mark.parametrize("model,twin", [(EB, False), (EB_TWIN, True)])
def test_lnl_batch_host_eb(model, twin):
    rng, t, flux = _lc(90)
    rows = synth.eb_rows(rng, 200, twin=twin, has_companion=True)
    got = lnl(model, t, flux, synth.SIGMA, tuple(rows))
    want = O.lnl_batch(model, t, flux, synth.SIGMA, rows)
    assert np.isposinf(want).any() or twin      # the secondary-depth rule is exercised
    _same(got, want, 1e-9)


def test_lnl_batch_host_scalar_broadcast_and_empty():
    """scalar columns broadcast like the reference's fixed-period call; n = 0 is a no-op"""
    rng, t, flux = _lc(64)
    rows = synth.tp_rows(rng, 50)
    cols = list(rows)
    cols[1] = 3.3                          # P_orb as a Python float, as lnZ_TTP passes it
    got = lnl(TP, t, flux, synth.SIGMA, cols)
    rows2 = rows.copy()
    rows2[1] = 3.3
    _same(got, O.lnl_batch(O.MODEL_TP, t, flux, synth.SIGMA, rows2), 1e-9)
    out = np.empty(0)
    assert _L.trx_lnl_batch_host(TP, 0, t, flux, len(t), synth.SIGMA, np.empty((10, 0)), 0,
                                 synth.EXPTIME, synth.NSAMPLES, out) == 0


def test_flux_grid_host():
    rng, t, _ = _lc(70)
    rows = synth.eb_rows(rng, 40, has_companion=True)
    grid, sec = np.empty((40, t.size)), np.empty(40)
    rc = _L.trx_flux_grid_host(EB, IS_HOST, t, t.size, rows, 40, synth.EXPTIME, synth.NSAMPLES,
                               grid, sec)
    assert rc == 0, _L.trx_last_error()
    wg, ws = O.flux_grid(O.MODEL_EB, t, rows, companion_is_host=True)
    assert np.max(np.abs(grid - wg)) < 5e-13
    assert np.max(np.abs(sec - ws.ravel())) < 5e-13


def test_log_mean_exp_host_reference_vectors():
    """the reference's own tests/test_log_mean_exp.py cases (tests/golden/numerics.npz)"""
    from helpers import gold
    g = gold("numerics.npz")
    for k in "abcdefg":
        x = np.ascontiguousarray(g["lme_in_" + k])
        out = np.empty(1)
        assert _L.trx_log_mean_exp_host(x, x.size, x.size, out) == 0
        want = g["lme_out_" + k][0]
        assert out[0] == want if not np.isfinite(want) else abs(out[0] - want) < 1e-12
    out = np.empty(1)
    assert _L.trx_log_mean_exp_host(np.zeros(5), 5, 6, out) == _lib.ERR_NTOTAL   # ValueError in the reference


def test_star_enqueue_equals_the_calls_one_by_one():
    """trx_star_enqueue as INTEGRATION.md section C binds it: the lnZ_* calls of one star -- here a planet and a
    binary scenario with their argument blocks built by triceratops_amd.fused -- handed to the library in ONE call on
    two streams, against the same blocks through trx_scenario_evidence (one call each, synchronous): the same records
    bit for bit (the draws depend on the seed in the block only)."""
    import triceratops_amd
    from helpers import gold
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    G = gold("lnz_cases.npz")
    base = (G["time"], G["flux"], float(G["sigma"][0]), 3.3, 0.82, 0.8, 5100.0, 0.0)
    triceratops_amd.set_sampling("device")
    saved = fused.TABLE_ROWS
    fused.TABLE_ROWS = 1
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.trx_star_enqueue.restype = ctypes.c_int
    L.trx_star_enqueue.argtypes = [ctypes.POINTER(fused.ScenarioArgs), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
    L.trx_scenario_evidence.restype = ctypes.c_int
    L.trx_scenario_evidence.argtypes = [ctypes.POINTER(fused.ScenarioArgs), ctypes.c_void_p]
    try:
        fused.begin_deferred(4)
        torch.manual_seed(3)
        pend = [ml.lnZ_TTP(*base, 200_000, True), ml.lnZ_TEB(*base, 200_000, True)]
        blocks = [b[0] for b in fused._tls.batch]            # the argument blocks fused collected (not yet enqueued)
        fused._tls.batch = []
        assert len(blocks) == 2 and all(isinstance(p, fused.Pending) for p in pend)
        n = len(blocks)
        calls = (fused.ScenarioArgs * n)(*blocks)
        recs = torch.zeros((n, fused.RECORD), dtype=torch.float64).pin_memory()
        streams = [torch.cuda.Stream() for _ in range(n)]
        for s_ in streams:
            _lib.wait_uploads(s_)                            # (the light curve went up asynchronously)
        outs = (ctypes.c_void_p * n)(*[recs[i].data_ptr() for i in range(n)])
        sts = (ctypes.c_void_p * n)(*[s_.cuda_stream for s_ in streams])
        done = ctypes.c_int(0)
        assert L.trx_star_enqueue(calls, n, outs, sts, ctypes.byref(done)) == 0, _L.trx_last_error()
        assert done.value == n
        for s_ in streams:
            s_.synchronize()
        together = recs.numpy().copy()
        # ... and one by one, synchronously
        for i, sa in enumerate(blocks):
            out = np.zeros(2 * fused.SCENARIO_OUT)
            flag = ctypes.c_int(-1)
            sa.out = out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            sa.out_flag = ctypes.pointer(flag)
            torch.cuda.synchronize()
            assert L.trx_scenario_evidence(ctypes.byref(sa), None) == 0, _L.trx_last_error()
            nbr = 1 if sa.draw.contents.planet else 2
            assert flag.value == 0 and together[i, 2 * fused.SCENARIO_OUT] == 0.0
            assert np.array_equal(out[:nbr * fused.SCENARIO_OUT], together[i, :nbr * fused.SCENARIO_OUT], equal_nan=True)
            ncol = 11 if nbr == 1 else 14
            assert np.isfinite(out[ncol]) and out[ncol + 1] > 1000          # lnZ, masked draws
        # argument checks: a null block list is refused, an empty one is a no-op
        assert L.trx_star_enqueue(None, 1, outs, sts, None) != 0
        assert L.trx_star_enqueue(calls, 0, outs, sts, ctypes.byref(done)) == 0 and done.value == 0
    finally:
        fused._tls.batch = []
        fused.end_deferred()
        fused.TABLE_ROWS = saved
        triceratops_amd.set_sampling("numpy")
