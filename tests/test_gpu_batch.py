"""BASELINE configs[3] and [4] on one GPU: a single-GPU shard of the 64-TOI batch (8 TOIs x 18
scenarios x N = 1e6 through calc_probs_many) in fp64 and in the mixed-precision mode
(TRX_FLAG_FP32_MODEL: fp32 flux model, fp64 orbit / chi^2 / log-mean-exp), and the fp32 kernel
against the CPU ORACLE (not against the fp64 kernel) on all 18 scenario families.

Mixed-precision tolerances (printed by the tests, stated in DESIGN.md section 9):
  flux      |fp32 - oracle| <= 2e-6 absolute
  chi^2/2   <= 1e-3 relative (worst on 50 % deep eclipses of nearby-star rows, chi^2 ~ 1e7; 0.05
            absolute for near-perfect fits), identical +inf pattern
  lnZ       <= 0.01 absolute on scenarios that carry probability (measured 7e-4), FPP <= 5e-5 (3.5e-6), NFPP <= 1e-7
"""
import os
import time

import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import GOLD
from oracle import oracle as O
from triceratops_amd import _lib, synth

pytestmark = pytest.mark.gpu

FLUX_ATOL_FP32 = 2e-6
H_RTOL_FP32 = 1e-3
H_ATOL_FP32 = 0.05


@pytest.mark.parametrize("n_time", [200, 2000])
def test_fp32_model_against_the_oracle(n_time):
    rng = np.random.default_rng(synth.SEED + 5)
    t = synth.time_grid(n_time)
    curve = O.flux_grid(O.MODEL_TP, t, synth.reference_tp_row())[0][0]
    flux = synth.noisy_light_curve(rng, curve)
    t_d, f_d = _lib.dev(t), _lib.dev(flux)
    n = 2000 if n_time == 200 else 200
    worst_h, worst_f = 0.0, 0.0
    for fam in synth.FAMILIES:
        name, model, is_host, _ = fam
        rows = synth.family_rows(rng, fam, n)
        flags = (_lib.FLAG_COMPANION_IS_HOST if is_host else 0) | _lib.FLAG_FP32_MODEL
        got = _lib.lnl_batch(model, flags, t_d, f_d, synth.SIGMA, _lib.dev(rows), synth.EXPTIME,
                             synth.NSAMPLES).cpu().numpy()
        want = O.lnl_batch(model, t, flux, synth.SIGMA, rows, companion_is_host=is_host)
        assert np.array_equal(np.isposinf(want), np.isposinf(got)), name
        fin = np.isfinite(want)
        err = np.abs(got[fin] - want[fin])
        assert np.all(err <= H_RTOL_FP32 * np.abs(want[fin]) + H_ATOL_FP32), (name, err.max())
        worst_h = max(worst_h, float(np.max(err / np.maximum(np.abs(want[fin]), 1.0))))
        g32, _ = _lib.flux_grid(model, flags, t_d, _lib.dev(rows[:, :64].copy()), synth.EXPTIME,
                                synth.NSAMPLES, False)
        g64, _ = O.flux_grid(model, t, rows[:, :64].copy(), companion_is_host=is_host)
        df = float(np.max(np.abs(g32.cpu().numpy() - g64)))
        assert df <= FLUX_ATOL_FP32, (name, df)
        worst_f = max(worst_f, df)
    print("fp32 model vs CPU oracle, %d points x 18 families x %d rows: max |dflux| = %.2e, "
          "max relative d(chi^2/2) = %.2e" % (n_time, n, worst_f, worst_h))


def _jobs(n_tois, N):
    return synth.toi_jobs(n_tois, n_time=200, N=N, seed=synth.SEED + 4,
                          trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                          contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))


def _batch(n_tois, N, sampling, precision, seed=11):
    import triceratops_amd
    triceratops_amd.set_sampling(sampling)
    triceratops_amd.set_precision(precision)
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        jobs = _jobs(n_tois, N)
        _lib.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = triceratops_amd.calc_probs_many(jobs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return out, dt, dict(_lib.STATS)
    finally:
        triceratops_amd.set_sampling("numpy")
        triceratops_amd.set_precision("fp64")


def test_config4_shard_fp64_vs_mixed_precision():
    """8 TOIs x 18 scenarios x N = 1e6 (one GPU's share of the 64-TOI batch), same draws (numpy's
    stream feeding the device pipeline) in both precisions"""
    N = 1_000_000
    a, dt64, st64 = _batch(8, N, "numpy-device", "fp64")
    b, dt32, st32 = _batch(8, N, "numpy-device", "fp32")
    assert st64 == st32 and st64["launches"] == 8 * 18      # 6 planet calls + 6 binary calls x 2 branches per TOI
    d_lnz, d_fpp, d_nfpp = 0.0, 0.0, 0.0
    for x, y in zip(a, b):
        assert len(x.lnZ) == 18 and x.FPP_degenerate is False
        fin = np.isfinite(x.lnZ)
        assert np.array_equal(fin, np.isfinite(y.lnZ))
        live = fin & (x.probs.prob.values > 1e-6)
        d_lnz = max(d_lnz, float(np.abs(x.lnZ[live] - y.lnZ[live]).max()))
        d_fpp, d_nfpp = max(d_fpp, abs(x.FPP - y.FPP)), max(d_nfpp, abs(x.NFPP - y.NFPP))
    print("config-4 shard, 8 TOIs x 18 x N=1e6, 200 points: fp64 %.2f s, fp32 model %.2f s; %d rows "
          "evaluated; max |dlnZ| = %.3g (scenarios with prob > 1e-6), max |dFPP| = %.3g, max |dNFPP| = %.3g"
          % (dt64, dt32, st64["rows"], d_lnz, d_fpp, d_nfpp))
    # (measured 7e-4, 3.5e-6, 1.3e-9: gates at ten times that)
    assert d_lnz < 0.01 and d_fpp < 5e-5 and d_nfpp < 1e-7


def test_config4_shard_device_sampling():
    """the same shard with the whole scenario on the GPU (what bench.py --mode batch times)"""
    out, dt, st = _batch(8, 1_000_000, "device", "fp64")
    out2, dt2, _ = _batch(8, 1_000_000, "device", "fp64")
    print("config-4 shard, device sampling: %.2f s first, %.2f s repeated, %d rows, %.3g cells/s"
          % (dt, dt2, st["rows"], st["cells"] / dt2))
    for x, y in zip(out, out2):
        assert x.FPP_degenerate is False and -1e-9 <= x.FPP <= 1.0 + 1e-9 and len(x.lnZ) == 18
        assert abs(x.FPP - y.FPP) < 1e-12          # same torch seed, same result


def test_threaded_units_give_the_same_results_as_one_thread():
    """set_threads(n): the scenarios of calc_probs_many evaluated side by side on n host threads /
    HIP streams; per-unit Philox keys make the result independent of n (bit for bit) and repeatable"""
    import triceratops_amd
    from triceratops_amd import sharding
    res = {}
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    try:
        for n_thr in (1, 3, 3):
            triceratops_amd.set_threads(n_thr)
            np.random.seed(21)
            torch.manual_seed(21)
            t0 = time.perf_counter()
            out = triceratops_amd.calc_probs_many(_jobs(6, 300_000))
            torch.cuda.synchronize()
            res.setdefault(n_thr, []).append((np.array([tg.lnZ for tg in out]), np.array([tg.FPP for tg in out]),
                                              time.perf_counter() - t0))
    finally:
        triceratops_amd.set_threads(1)
        sharding.per_unit_seed = False
        triceratops_amd.set_sampling("numpy")
    one, thr_a, thr_b = res[1][0], res[3][0], res[3][1]
    print("6 TOIs x 18 x N=3e5: 1 thread %.3f s, 3 threads %.3f s" % (one[2], thr_b[2]))
    assert np.array_equal(one[0], thr_a[0], equal_nan=True) and np.array_equal(one[1], thr_a[1])
    assert np.array_equal(thr_a[0], thr_b[0], equal_nan=True)


def test_one_threads_per_call_choices_do_not_reach_another_threads_calls():
    """rounds 1-5 exported process-wide switches from the production library, and a switch
    flipped by one host thread changed what another thread's in-flight calc_probs enqueued.  The production library has
    none any more (tests/test_abi.py): what a caller may choose is a TRX_FLAG_* bit of ITS call.  One thread loops over
    likelihood calls with every result-neutral flag set -- all sub-exposures, no stencil, evaluation counts, every
    excluded row evaluated -- and calc_probs with the bounded evaluation off, while another runs calc_probs_many on
    the defaults: each thread's results equal what it gets alone, bit for bit."""
    import threading
    import triceratops_amd
    from triceratops_amd import fused, sharding
    assert _lib.lib().trx_testing is False                     # the production library
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    rng = np.random.default_rng(8)
    t = synth.time_grid(2000)
    t_d = _lib.dev(t)
    f_d = _lib.dev(1.0 + rng.normal(0, synth.SIGMA, 2000))
    rows = _lib.dev(synth.eb_rows(rng, 4000, False, True))
    odd = _lib.FLAG_ALL_SUBEXPOSURES | _lib.FLAG_EVALUATE_EXCLUDED

    def noisy_neighbour(out):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            _lib.wait_uploads(st)
            for _ in range(out["reps"]):
                h = _lib.lnl_batch(1, odd, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
                g, _ = _lib.flux_grid(1, _lib.FLAG_NO_STENCIL, t_d, rows[:, :300].contiguous(), synth.EXPTIME, 20, False)
                c, _ = _lib.flux_grid(1, _lib.FLAG_COUNT_EVALUATIONS, t_d, rows[:, :300].contiguous(), synth.EXPTIME, 20, False)
            st.synchronize()
        out["h"], out["g"], out["c"] = h.cpu().numpy(), g.cpu().numpy(), c.cpu().numpy()

    def tables():
        np.random.seed(3)
        torch.manual_seed(3)
        out = triceratops_amd.calc_probs_many(_jobs(4, 200_000))
        return np.array([tg.lnZ for tg in out]), np.array([tg.FPP for tg in out])

    try:
        alone = {"reps": 1}
        noisy_neighbour(alone)
        want = tables()
        assert alone["c"].max() == 20.0 and alone["c"].min() == 0.0       # counts, not fluxes
        busy = {"reps": 40}
        th = threading.Thread(target=noisy_neighbour, args=(busy,))
        th.start()
        got = [tables() for _ in range(3)]
        th.join()
        for lnz, fpp in got:
            assert np.array_equal(lnz, want[0], equal_nan=True) and np.array_equal(fpp, want[1])
        for k in ("h", "g", "c"):
            assert np.array_equal(busy[k], alone[k], equal_nan=True)
        # ... and the per-call form of "no bounded evaluation" gives the full evaluation's records
        _lib.reset_stats()
        _lib.check(_lib.lib().trx_pruned_rows(None, 1))
        triceratops_amd.set_full_evaluation(True)
        full = tables()
        n = ctypes.c_ulonglong(0)
        _lib.check(_lib.lib().trx_pruned_rows(ctypes.byref(n), 1))
        assert n.value == 0                                         # nothing abandoned
        assert np.allclose(full[0], want[0], rtol=0, atol=1e-9, equal_nan=True) and np.allclose(full[1], want[1], atol=1e-12)
        triceratops_amd.set_full_evaluation(False)
        tables()
        _lib.check(_lib.lib().trx_pruned_rows(ctypes.byref(n), 1))
        assert n.value > 0
    finally:
        triceratops_amd.set_full_evaluation(False)
        sharding.per_unit_seed = False
        triceratops_amd.set_sampling("numpy")


def test_native_scenario_call_equals_the_torch_operator_path():
    """trx_scenario_enqueue (draws -> compaction -> likelihood -> evidence -> best draw in one library
    call, no host sync) against the chain of torch operators around trx_draw_scenario / trx_lnz_scenario on
    the same Philox keys, on all 18 scenarios of several TOIs, with a contrast curve, in fp64 and in the
    mixed-precision mode.  With every row evaluated to the end (trx_set_bounded_evaluation(0)): every lnZ, every best-fit column and FPP / NFPP bit for bit.  With the
    bounded evaluation (mode 2, the default): the same best draws, lnZ to 1e-12 (the probe cells of a row are summed
    first: another order of the same terms)."""
    import triceratops_amd
    from triceratops_amd import fused, sharding
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    cols = ("M_s", "R_s", "P_orb", "inc", "b", "R_p", "ecc", "w", "M_EB", "R_EB")
    attrs = ("u1", "u2", "fluxratio_EB", "fluxratio_comp", "star_num")
    L = _lib.lib()
    try:
        for precision in ("fp64", "fp32"):
            triceratops_amd.set_precision(precision)
            got = {}
            for mode in ("native", "native-unbounded", "torch"):
                fused.NATIVE = mode != "torch"
                L.trx_set_bounded_evaluation(0 if mode == "native-unbounded" else 2)
                np.random.seed(5)
                torch.manual_seed(5)
                _lib.reset_stats()
                out = triceratops_amd.calc_probs_many(_jobs(4, 200_000))
                got[mode] = (out, dict(_lib.STATS))
            for x, y, z in zip(got["native-unbounded"][0], got["torch"][0], got["native"][0]):
                assert x.FPP == y.FPP and x.NFPP == y.NFPP
                assert np.array_equal(x.lnZ, y.lnZ, equal_nan=True) and np.array_equal(x.probs["prob"].values, y.probs["prob"].values)
                fin = np.isfinite(y.lnZ)
                assert np.array_equal(fin, np.isfinite(z.lnZ))
                assert np.allclose(z.lnZ[fin], y.lnZ[fin], rtol=1e-12, atol=0)
                assert abs(z.FPP - y.FPP) < 1e-12 and abs(z.NFPP - y.NFPP) < 1e-12
                for c in cols:
                    assert np.array_equal(x.probs[c].values, y.probs[c].values, equal_nan=True), (precision, c)
                    assert np.array_equal(z.probs[c].values, y.probs[c].values, equal_nan=True), (precision, c)
                for c in attrs:
                    assert np.array_equal(np.asarray(getattr(x, c)), np.asarray(getattr(y, c)), equal_nan=True), (precision, c)
                    assert np.array_equal(np.asarray(getattr(z, c)), np.asarray(getattr(y, c)), equal_nan=True), (precision, c)
            assert got["native"][1]["rows"] == got["torch"][1]["rows"] > 0
            assert got["native"][1]["cells"] == got["torch"][1]["cells"]
    finally:
        fused.NATIVE = True
        L.trx_set_bounded_evaluation(2)          # (the library's default)
        sharding.per_unit_seed = False
        triceratops_amd.set_precision("fp64")
        triceratops_amd.set_sampling("numpy")


def test_bounded_evaluation_leaves_no_row_unwritten_whatever_the_row_count():
    """The bounded evaluation of short light curves is three passes (pilot rows, probe pass, the rows left alive)
    whose grids are sized for an upper bound of a row count that only the device knows; a workgroup beyond the
    batches leaves at once, by a rule that must use the SAME rows-per-wave as the pass itself (round 4's first
    version did not for 30 000-34 000 masked draws: tests/test_toi465.py::test_blend_bounded_evaluation_equals_the_
    full_one_on_one_and_six_streams is the run that showed it).  Here: N swept so that the masked counts of the 18
    scenarios of two synthetic TOIs cross the thresholds of the rows-per-wave rule, with the chi^2 arrays poisoned
    before every call (trx_set_debug_poison: an unwritten row reads as a perfect fit); bounded against unbounded:
    same best draws, lnZ to 1e-12.  (Verified to fail on a build with the old rule.)"""
    import triceratops_amd
    from triceratops_amd import sharding
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    L = _lib.lib()
    L.trx_set_debug_poison(1)
    try:
        rng = np.random.default_rng(3)
        for N in (60_000, 250_000, 290_000, 330_000, 370_000, 420_000, 480_000, 560_000):
            # with the TOIs' transits, and with pure noise in their place (no draw stands out: the pilot finds
            # nothing to abandon and the third pass takes every row behind it -- the case that went wrong)
            for signal in (True, False):
                got = {}
                for mode in (0, 2):
                    L.trx_set_bounded_evaluation(mode)
                    np.random.seed(9)
                    torch.manual_seed(9)
                    jobs = _jobs(2, N)
                    if not signal:
                        for _, kw in jobs:
                            kw["flux_0"] = 1.0 + np.random.default_rng(4).normal(0.0, kw["flux_err_0"], kw["time"].size)
                    got[mode] = triceratops_amd.calc_probs_many(jobs)
                for x, z in zip(got[0], got[2]):
                    fin = np.isfinite(x.lnZ)
                    assert np.array_equal(fin, np.isfinite(z.lnZ)), (N, signal)
                    assert np.allclose(z.lnZ[fin], x.lnZ[fin], rtol=1e-12, atol=0), (N, signal, np.abs(z.lnZ[fin] - x.lnZ[fin]).max())
                    for c in ("P_orb", "inc", "R_p", "ecc", "w", "M_EB", "R_EB"):
                        assert np.array_equal(x.probs[c].values, z.probs[c].values, equal_nan=True), (N, signal, c)
    finally:
        L.trx_set_debug_poison(0)
        L.trx_set_bounded_evaluation(2)
        sharding.per_unit_seed = False
        triceratops_amd.set_sampling("numpy")


@pytest.mark.parametrize("n_time", [20, 40, 47, 48, 63])
def test_bounded_evaluation_on_very_short_light_curves(n_time):
    """Below 48 stamps there is nothing to probe and the launch is evaluated in full.  Until profiles/fuzz_bounded.py
    ran, such a launch still went through the passes: the probe pass declined (stride 1) while the third pass waited
    for the probe pass's list, and the rows behind the pilot were never written when the pilot's verdict was
    "probing pays" (every n_time < 48 with a clear signal).  Poisoned chi^2 arrays, bounded against full, both sides
    of the threshold.  (Fails on the library of the commit before.)"""
    import triceratops_amd
    from triceratops_amd import sharding
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    L = _lib.lib()
    L.trx_set_debug_poison(1)
    try:
        got = {}
        for mode in (0, 2):
            L.trx_set_bounded_evaluation(mode)
            np.random.seed(5)
            torch.manual_seed(5)
            jobs = synth.toi_jobs(2, n_time=n_time, N=130_000, seed=synth.SEED + 9,
                                  trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                                  contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
            got[mode] = triceratops_amd.calc_probs_many(jobs)
        for x, z in zip(got[0], got[2]):
            fin = np.isfinite(x.lnZ)
            assert np.array_equal(fin, np.isfinite(z.lnZ))
            assert np.allclose(z.lnZ[fin], x.lnZ[fin], rtol=1e-12, atol=0), np.abs(z.lnZ[fin] - x.lnZ[fin]).max()
            for c in ("P_orb", "inc", "R_p", "ecc", "w", "M_EB", "R_EB"):
                assert np.array_equal(x.probs[c].values, z.probs[c].values, equal_nan=True), c
    finally:
        L.trx_set_debug_poison(0)
        L.trx_set_bounded_evaluation(2)
        sharding.per_unit_seed = False
        triceratops_amd.set_sampling("numpy")


def test_launches_whose_batches_exceed_the_grid_caps():
    """cells_kernel's grids are capped when the row count lives on the device (1280 workgroups for the passes of the
    bounded evaluation, 5120 for a full evaluation: launch_cells) and the waves stride over the batches beyond the
    cap.  N = 3e6 draws on a 100-point light curve leaves ~3e5 masked rows = 12 000 workgroups' worth of batches,
    beyond both caps: the native call without the bounded evaluation against the torch-operator chain (whose
    launches know their row count on the host: exact grids, no cap) bit for bit, the bounded one to 1e-12 with the
    same best draws."""
    import triceratops_amd
    from triceratops_amd import fused, sharding
    triceratops_amd.set_sampling("device")
    sharding.per_unit_seed = True
    L = _lib.lib()
    L.trx_set_debug_poison(1)
    try:
        got = {}
        for mode in ("native", "native-unbounded", "torch"):
            fused.NATIVE = mode != "torch"
            L.trx_set_bounded_evaluation(0 if mode == "native-unbounded" else 2)
            np.random.seed(6)
            torch.manual_seed(6)
            jobs = synth.toi_jobs(1, n_time=100, N=3_000_000, seed=synth.SEED + 5,
                                  trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"), contrast_curve_file=None)
            got[mode] = triceratops_amd.calc_probs_many(jobs)[0]
        x, y, z = got["native-unbounded"], got["torch"], got["native"]
        assert np.array_equal(x.lnZ, y.lnZ, equal_nan=True) and x.FPP == y.FPP
        fin = np.isfinite(y.lnZ)
        assert np.array_equal(fin, np.isfinite(z.lnZ))
        assert np.allclose(z.lnZ[fin], y.lnZ[fin], rtol=1e-12, atol=0)
        for c in ("P_orb", "inc", "R_p", "ecc", "w", "M_EB", "R_EB"):
            assert np.array_equal(x.probs[c].values, y.probs[c].values, equal_nan=True), c
            assert np.array_equal(z.probs[c].values, y.probs[c].values, equal_nan=True), c
    finally:
        fused.NATIVE = True
        L.trx_set_debug_poison(0)
        L.trx_set_bounded_evaluation(2)
        sharding.per_unit_seed = False
        triceratops_amd.set_sampling("numpy")
