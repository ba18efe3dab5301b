"""Host-side mirror of the reference (priors, funcs, _numerics, marginal_likelihoods, calc_probs)
against golden vectors produced by the imported reference (tests/golden/make_golden.py).

These run on a CPU-only box: the device entry points are replaced by an oracle-backed stand-in
(tests/helpers.py) so that what is under test is the host logic -- draw order on the global
numpy stream, derived columns, masks, priors, evidence bookkeeping, best-fit tables.  The real
kernels are exercised by the -m gpu tests with the same golden files.
"""
import os

import numpy as np
import pytest

from helpers import GOLD, gold, install_cpu_device_fakes
from triceratops_amd import funcs, priors


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


# ---------------------------------------------------------------------------------------
def test_samplers_bit_identical_to_reference():
    g = gold("priors_funcs.npz")
    x = g["uniforms"]
    for M in (1.3, 1.0, 0.7, 0.3, 0.25, 0.1, 0.05):
        assert _same(priors.sample_q(x.copy(), M), g["sample_q_%g" % M]), M
        assert _same(priors.sample_q_companion(x.copy(), M), g["sample_qc_%g" % M]), M
    assert _same(priors.sample_rp(x.copy(), g["rp_masses"], False), g["sample_rp"])
    assert _same(priors.sample_rp(x.copy(), g["rp_masses"], True), g["sample_rp_flat"])
    assert _same(priors.sample_inc(x.copy()), g["sample_inc"])
    assert _same(priors.sample_w(x.copy()), g["sample_w"])
    for planet, P in ((True, 3.0), (False, 3.0), (False, 20.0)):
        np.random.seed(77)
        assert _same(priors.sample_ecc(x.copy(), planet, P), g["sample_ecc_%d_%g" % (planet, P)])
    # samplers do not touch the caller's array (the reference mutates it in place)
    y = x.copy()
    priors.sample_q(y, 0.7)
    priors.sample_rp(y, g["rp_masses"], False)
    assert _same(y, x)


def test_stellar_and_flux_relations_bit_identical():
    g = gold("priors_funcs.npz")
    m = g["sr_masses"]
    r, t = funcs.stellar_relations(m, np.full(m.size, 1.1), np.full(m.size, 6100.0))
    assert _same(r, g["sr_radii"]) and _same(t, g["sr_teffs"])
    for band in ("TESS", "Vis", "J", "H", "K"):
        assert _same(funcs.flux_relation(m, band), g["flux_relation_" + band])
    fl, fe = funcs.renorm_flux(np.array([1.0, 0.999, 0.9985]), 5e-4, 0.37)
    assert _same(fl, g["renorm_flux"]) and fe == g["renorm_err"][0]


def test_companion_priors_bit_identical():
    g = gold("priors_funcs.npz")
    dm = g["prior_dmags"]
    seps, cons = funcs.file_to_contrast_curve(os.path.join(GOLD, "contrast_curve_synth.csv"))
    for M in (1.25, 0.8):
        for plx in (12.5, np.nan):
            tag = "%g_%s" % (M, "nan" if np.isnan(plx) else "%g" % plx)
            assert _same(priors.lnprior_bound_TP(M, plx, dm, seps, cons), g["bound_TP_" + tag])
            assert _same(priors.lnprior_bound_EB(M, plx, dm, seps, cons), g["bound_EB_" + tag])
            assert _same(priors.lnprior_bound_TP(M, plx, dm, np.array([2.2]), np.array([1.0])),
                         g["bound_TP_nocc_" + tag])
    assert _same(priors.lnprior_background(391, dm, seps, cons), g["background"])
    # natural log of (N/0.1)(1/3600)^2 sep^2 (reference tests/test_background_prior_log_base.py)
    sep = np.interp(dm, cons, seps)
    assert np.allclose(priors.lnprior_background(391, dm, seps, cons),
                       np.log((391 / 0.1) * (1 / 3600) ** 2 * sep ** 2), rtol=1e-15)


def test_normalize_probabilities_and_lme_guard(monkeypatch):
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd._numerics import _log_mean_exp, _normalize_probabilities
    g = gold("numerics.npz")
    for k in ("ok", "allneg", "anom"):
        p, st = _normalize_probabilities(g["norm_in_" + k])
        assert st == str(g["norm_status_" + k][0]) and np.allclose(p, g["norm_out_" + k], atol=1e-16)
    for k in "abcdefg":
        x, want = g["lme_in_" + k], g["lme_out_" + k][0]
        got = _log_mean_exp(x, N_total=x.size)
        assert (got == want) if not np.isfinite(want) else abs(got - want) < 1e-12
    with pytest.raises(ValueError):
        _log_mean_exp(np.zeros(5), N_total=4)


# ---------------------------------------------------------------------------------------
def _star(g):
    return dict(zip(("M_s", "R_s", "Teff", "Z", "plx", "Tmag", "Jmag", "Hmag", "Kmag"), g["star"]))


def _call(ml, name, g, P, N, parallel, cc, filt):
    s = _star(g)
    t, f, sigma = g["time"], g["flux"], float(g["sigma"][0])
    tri = os.path.join(GOLD, "trilegal_synth.csv")
    base = (t, f, sigma, P, float(s["M_s"]), float(s["R_s"]), float(s["Teff"]))
    fn = getattr(ml, "lnZ_" + name)
    if name in ("TTP", "TEB"):
        return fn(*base, 0.0, N, parallel)
    if name in ("PTP", "PEB", "STP", "SEB"):
        return fn(*base, 0.0, float(s["plx"]), cc, filt, N, parallel)
    mags = tuple(float(s[k]) for k in ("Tmag", "Jmag", "Hmag", "Kmag"))
    if name in ("DTP", "DEB"):
        return fn(*base, 0.0, *mags, tri, cc, filt, N, parallel)
    return fn(*base, *mags, tri, cc, filt, N, parallel)


CASES = [str(c) for c in gold("lnz_cases.npz")["cases"]]


@pytest.mark.parametrize("case", CASES)
def test_lnz_functions_reproduce_reference_draw_for_draw(case, monkeypatch):
    """same np.random.seed -> same per-draw log-weights, lnZ and best-fit tables as the reference
    (whose model calls went through the oracle's QuadraticModel)"""
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd import marginal_likelihoods as ml
    g = gold("lnz_cases.npz")
    name, variant = case.split("_")
    P = [2.5, 4.0] if variant == "range" else 3.3
    cc = os.path.join(GOLD, "contrast_curve_synth.csv") if variant == "ccJ" else None
    filt = "J" if variant == "ccJ" else "TESS"
    parallel = variant != "serial"
    N = int(g["N"][0]) if parallel else 300
    captured = []
    real = ml._lib.lnz_scenario

    def spy(model, flags, t, f, sigma, params, exptime, nsamples, lnprior, n_total, lnsigma):
        h, lnz = real(model, flags, t, f, sigma, params, exptime, nsamples, lnprior, n_total, lnsigma)
        captured.append((params.numpy().copy(), h.numpy().copy()))
        return h, lnz

    monkeypatch.setattr(ml._lib, "lnz_scenario", spy)
    np.random.seed(int(g[case + "_seed"][0]))
    res = _call(ml, name, g, P, N, parallel, cc, filt)
    dicts = res if isinstance(res, tuple) else (res,)
    assert len(dicts) == int(g[case + "_nres"][0])
    for i, d in enumerate(dicts):
        want_lnz = g["%s_lnZ%d" % (case, i)][0]
        if np.isfinite(want_lnz):
            assert abs(d["lnZ"] - want_lnz) < 1e-10, (case, i, d["lnZ"], want_lnz)
        else:
            assert d["lnZ"] == want_lnz
        assert isinstance(d["lnZ"], float)
        logw = g["%s_logw%d" % (case, i)]
        n_fin = int(np.isfinite(logw).sum())
        for k in ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB",
                  "R_EB", "fluxratio_EB", "fluxratio_comp"):
            want = g["%s_res%d_%s" % (case, i, k)]
            assert d[k].shape == (100,)
            # rows beyond the finite draws are ties at -inf whose order is arbitrary
            top = min(n_fin, 100) if k not in ("M_s", "R_s", "u1", "u2") or d[k].std() > 0 else 100
            assert np.allclose(d[k][:top], want[:top], rtol=1e-13, atol=0), (case, i, k)
    if parallel:
        # the parameter blocks handed to the kernels are the reference's, bit for bit
        for i, (block, h) in enumerate(captured):
            want_block = g["%s_call%d_block" % (case, i)]
            assert _same(block, want_block), (case, i)
            want_h = g["%s_call%d_out" % (case, i)]
            fin = np.isfinite(want_h)
            assert np.array_equal(np.isposinf(h), np.isposinf(want_h))
            assert np.allclose(h[fin], want_h[fin], rtol=1e-12, atol=0)


def test_toi1228_config1(monkeypatch):
    """BASELINE configs[0]: TOI-1228 folded light curve, TP scenario, N = 1e4"""
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd import marginal_likelihoods as ml
    g = gold("toi1228_ttp.npz")
    M, R, Teff, Z = (float(v) for v in g["star"])
    np.random.seed(int(g["seed"][0]))
    res = ml.lnZ_TTP(g["time"], g["flux"], float(g["sigma"][0]), float(g["P_orb"][0]), M, R, Teff,
                     Z, int(g["N"][0]), True)
    assert abs(res["lnZ"] - g["lnZ"][0]) < 1e-10
    for k in ("P_orb", "inc", "b", "R_p", "ecc", "argp"):
        assert np.allclose(res[k], g["res_" + k], rtol=1e-13)


def test_numpy_scalar_period_takes_the_range_branch(monkeypatch):
    """`type(P_orb) not in [float, int]` (marginal_likelihoods.py:67): a numpy float is a range"""
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd import marginal_likelihoods as ml
    g = gold("lnz_cases.npz")
    with pytest.raises((IndexError, TypeError)):
        ml.lnZ_TTP(g["time"], g["flux"], 6e-4, np.float64(3.3), 0.8, 0.8, 5100.0, 0.0, 50, True)


def test_missing_ldc_combination_raises_like_the_reference(monkeypatch):
    """SEB allows rounded companion Teff up to 13000 K, past the 10000 K end of the grid
    (marginal_likelihoods.py:1181): .item() on the empty match raises ValueError"""
    from triceratops_amd import marginal_likelihoods as ml
    tab = ml._ldc("TESS")
    u1, u2 = tab.companions(0.0, np.array([5000.0, 9900.0]), np.array([4.4, 4.1]), 13000)
    assert u1.shape == (2,)
    with pytest.raises(ValueError):
        tab.companions(0.0, np.array([12400.0]), np.array([4.0]), 13000)


EXTRA = [str(c) for c in gold("lnz_extra.npz")["cases"]]


@pytest.mark.parametrize("case", EXTRA)
def test_unused_lnz_functions_match_reference(case, monkeypatch):
    """lnZ_NTP/NEB_unknown and _evolved: exported by the reference, never called by calc_probs
    (SURVEY 8 row a9); quirks kept (one dict for an empty population, no u1/u2 keys, twin = host copy)"""
    from helpers import call_extra, check_extra
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd import marginal_likelihoods as ml
    g = gold("lnz_extra.npz")
    np.random.seed(int(g[case + "_seed"][0]))
    check_extra(call_extra(ml, case, g), case, g, 1e-10)


def test_constant_period_semi_major_axis_shortcut_is_bit_identical():
    from triceratops_amd import marginal_likelihoods as ml
    from triceratops_amd.constants import G, Msun, pi
    rng = np.random.default_rng(0)
    for M, P in zip(rng.uniform(0.1, 3.0, 40), rng.uniform(0.3, 500.0, 40)):
        P_arr = np.full(1000, P)
        want = ((G * M * Msun) / (4 * pi ** 2) * (P_arr * 86400) ** 2) ** (1 / 3)
        assert np.array_equal(ml._sma(M, P_arr), want)
    P_var = rng.uniform(1.0, 5.0, 1000)                     # varying period: the general expression
    assert np.array_equal(ml._sma(1.0, P_var), ((G * 1.0 * Msun) / (4 * pi ** 2) * (P_var * 86400) ** 2) ** (1 / 3))
    M_var = rng.uniform(0.5, 2.0, 1000)
    assert np.array_equal(ml._sma(M_var, np.full(1000, 3.3)),
                          ((G * M_var * Msun) / (4 * pi ** 2) * (np.full(1000, 3.3) * 86400) ** 2) ** (1 / 3))


def test_default_sampling_mode_is_the_device_path():
    """a user who drops the package in gets the fast mode; "numpy" is the validation mode (conftest.py selects it)"""
    import subprocess
    import sys
    code = ("import triceratops_amd, triceratops_amd.marginal_likelihoods as ml;"
            "assert ml.DEFAULT_SAMPLING == 'device' and triceratops_amd.get_sampling() == 'device'")
    env = {k: v for k, v in __import__("os").environ.items() if k != "TRX_SAMPLING"}
    assert subprocess.run([sys.executable, "-c", code], cwd=__import__("os").path.dirname(__import__("os").path.dirname(
        __import__("os").path.abspath(__file__))), env=env).returncode == 0


def test_field_star_limb_darkening_lookup_equals_the_dense_comparison():
    """_LdcTable.field_stars looks a star up in its (Teff, logg) cell; the reference compares every star with
    every table row (marginal_likelihoods.py:1913-1924).  Same rows, ties, NaNs and failures."""
    from triceratops_amd import marginal_likelihoods as ml

    def dense(tab, Teffs, loggs, Zs):
        t = tab.Teffs[np.argmin(np.abs(tab.Teffs[None, :] - Teffs[:, None]), axis=1)]
        g = tab.loggs[np.argmin(np.abs(tab.loggs[None, :] - loggs[:, None]), axis=1)]
        cell = (tab.Teffs[None, :] == t[:, None]) & (tab.loggs[None, :] == g[:, None])
        dz = np.where(cell, np.abs(tab.Zs[None, :] - Zs[:, None]), np.inf)
        z = tab.Zs[np.argmin(dz, axis=1)]
        row = cell & (tab.Zs[None, :] == z[:, None])
        if not np.all(row.sum(axis=1) == 1):
            raise ValueError("size")
        idx = np.argmax(row, axis=1)
        return tab.u1s[idx], tab.u2s[idx]

    rng = np.random.default_rng(5)
    for mission in ("TESS", "Kepler"):
        tab = ml._ldc(mission)
        # stars on cells the grid has (any Teff/logg pair of an existing row, jittered), mid-points (ties), NaN Z
        j = rng.integers(0, tab.Teffs.size, 3000)
        Teffs = tab.Teffs[j] + rng.uniform(-100, 100, j.size)
        loggs = tab.loggs[j] + rng.uniform(-0.2, 0.2, j.size)
        Zs = rng.uniform(-6, 2, j.size)
        Teffs[:200] = tab.Teffs[j[:200]] + 125.0          # exactly between two nodes where the grid steps by 250 K
        loggs[100:300] = tab.loggs[j[100:300]] + 0.25
        Zs[300:320] = np.nan
        Zs[320:400] = np.round(Zs[320:400] * 2) / 2 + 0.25   # between two metallicities
        ok = np.ones(j.size, dtype=bool)
        for i in range(j.size):
            try:
                dense(tab, Teffs[i:i + 1], loggs[i:i + 1], Zs[i:i + 1])
            except ValueError:
                ok[i] = False
        assert ok.sum() > 2000
        a = dense(tab, Teffs[ok], loggs[ok], Zs[ok])
        b = tab.field_stars(Teffs[ok], loggs[ok], Zs[ok])
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)
        for i in np.flatnonzero(~ok)[:20]:
            with pytest.raises(ValueError):
                tab.field_stars(Teffs[i:i + 1], loggs[i:i + 1], Zs[i:i + 1])


def test_eccentricity_quantile_table_of_the_draw_kernel():
    """csrc/trx_ecc_icdf.inc (generated by profiles/r06/make_ecc_icdf.py) is what the draw kernel samples the planets'
    Beta(0.867, 3.030) eccentricities from (priors.py:146-148; ecc_from_uniform in csrc/trx_draw.hip).  The committed
    table, read back from the source file and interpolated the way the kernel does it -- fp32, linear in u^(1/a) below
    the median and in (1 - u)^(1/b) above --, is scipy's quantile to 2e-6 everywhere, monotone, and ends at 0 and 1."""
    import re
    from scipy import special
    src = open(os.path.join(os.path.dirname(__file__), "..", "triceratops_amd", "csrc", "trx_ecc_icdf.inc")).read()
    M = int(re.search(r"kEccIcdfN = (\d+)", src).group(1))
    wA, vB = (np.float32(re.search(r"%s = ([0-9.e+-]+)f" % k, src).group(1)) for k in ("kEccIcdfWA", "kEccIcdfVB"))
    tabs = {}
    for name in ("kEccIcdfLo", "kEccIcdfHi"):
        body = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, src, re.S).group(1)
        tabs[name] = np.array([float(x.rstrip("f")) for x in body.replace("\n", " ").split(",") if x.strip()], dtype=np.float32)
        assert tabs[name].size == M + 1
    lo, hi = tabs["kEccIcdfLo"], tabs["kEccIcdfHi"]
    assert lo[0] == 0.0 and hi[0] == 1.0 and abs(float(lo[-1]) - float(hi[-1])) < 1e-7
    assert np.all(np.diff(lo) > 0) and np.all(np.diff(hi) < 0)
    a, b = 0.867, 3.030
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.random(400000), [0.0, 0.5, 1 - 2.0 ** -53, 1e-12, 0.4999999, 0.5000001]])
    is_lo = u < 0.5
    x = np.where(is_lo, u, 1.0 - u).astype(np.float32)
    pw = np.where(is_lo, np.float32(1 / a), np.float32(1 / b))
    with np.errstate(divide="ignore"):
        x = np.exp2(pw * np.log2(x)).astype(np.float32)                  # 0 -> 0, as v_log_f32 / v_exp_f32 give
    f = x * np.where(is_lo, np.float32(M) / wA, np.float32(M) / vB)
    j = np.clip(f.astype(np.int32), 0, M - 1)
    fr = f - j
    e0 = np.where(is_lo, lo[j], hi[j])
    e1 = np.where(is_lo, lo[j + 1], hi[j + 1])
    ecc = fr * (e1 - e0) + e0
    assert np.abs(ecc - special.betaincinv(a, b, u)).max() < 2e-6
    assert ecc.min() >= 0.0 and ecc.max() <= 1.0
