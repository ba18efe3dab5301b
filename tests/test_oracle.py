"""Pinning the CPU oracle (oracle/trx_oracle.c).

 * _log_mean_exp / _normalize_probabilities: against values produced by the imported reference
   module (tests/golden/numerics.npz) and the exact known answers the reference's own
   tests/test_log_mean_exp.py asserts.
 * unit conversion, radius-ratio rule, dilution, secondary depth, chi^2/2, +inf rule: against the
   blocks and outputs captured from the reference's own likelihoods.py while it ran on top of the
   oracle's QuadraticModel (tests/golden/lnz_cases.npz, made by tests/golden/make_golden.py).
 * the Mandel-Agol / Kepler arithmetic (parity unpinned against pytransit, see oracle header):
   against closed forms and an independent arbitrary-precision quadrature of the limb-darkened
   disk.
"""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import gold


# ---------------------------------------------------------------------------------------
# _numerics
def test_lme_matches_reference_outputs():
    g = gold("numerics.npz")
    for k in "abcdefg":
        x, want = g["lme_in_" + k], g["lme_out_" + k][0]
        got = O.log_mean_exp(x, x.size)
        assert (got == want) if not np.isfinite(want) else abs(got - want) < 1e-12, k


def test_lme_known_answers_of_reference_tests():
    # values asserted in the reference's tests/test_log_mean_exp.py (24-270)
    assert abs(O.log_mean_exp(np.full(1000, -2000.0), 1000) - (-2000.0)) < 1e-10
    x = np.array([-1001.0, -1002.0, -1003.0, -1004.0, -1005.0] + [-np.inf] * 5)
    m = x[:5].max()
    want = m + np.log(np.sum(np.exp(x[:5] - m))) - np.log(10)
    assert abs(O.log_mean_exp(x, 10) - want) < 1e-12
    assert abs(O.log_mean_exp(np.array([-1.0] + [-np.inf] * 9), 10) - (-1 - np.log(10))) < 1e-14
    assert abs(O.log_mean_exp(np.array([-1.0, np.nan, -np.inf, np.nan, -np.inf]), 5) - (-1 - np.log(5))) < 1e-14
    for n in (1, 10, 100000):
        assert O.log_mean_exp(np.full(n, -np.inf), n) == -np.inf
    assert O.log_mean_exp(np.array([-1.0, np.inf]), 2) == np.inf
    assert O.log_mean_exp(np.array([-3.5]), 1) == -3.5
    with pytest.raises(ValueError):
        O.log_mean_exp(np.zeros(4), 3)


def test_normalize_matches_reference_outputs():
    g = gold("numerics.npz")
    for k in ("ok", "allneg", "anom"):
        p, st = O.normalize_probabilities(g["norm_in_" + k])
        assert st == str(g["norm_status_" + k][0])
        assert np.allclose(p, g["norm_out_" + k], rtol=0, atol=1e-15)
    p, st = O.normalize_probabilities(np.array([0.0, 1.0]))
    assert st == "ok" and abs(p[1] / p[0] - np.e) < 1e-12
    p, _ = O.normalize_probabilities(np.array([-np.inf, -5.0]))
    assert p[0] == 0.0 and p[1] == 1.0


# ---------------------------------------------------------------------------------------
# likelihoods.py, as driven by the reference itself
def _calls(g):
    for case in g["cases"]:
        i = 0
        while "%s_call%d_name" % (case, i) in g.files:
            yield case, i
            i += 1


def test_lnl_matches_reference_likelihoods_on_captured_blocks():
    g = gold("lnz_cases.npz")
    t, f, sigma = g["time"], g["flux"], float(g["sigma"][0])
    model = {"lnL_TP_p": O.MODEL_TP, "lnL_EB_p": O.MODEL_EB, "lnL_EB_twin_p": O.MODEL_EB_TWIN}
    n_checked = 0
    for case, i in _calls(g):
        name = str(g["%s_call%d_name" % (case, i)][0])
        block = g["%s_call%d_block" % (case, i)]
        want = g["%s_call%d_out" % (case, i)]
        is_host = bool(g["%s_call%d_is_host" % (case, i)][0])
        if block.shape[1] == 0:
            continue
        got = O.lnl_batch(model[name], t, f, sigma, block, companion_is_host=is_host)
        assert np.array_equal(np.isposinf(got), np.isposinf(want)), (case, i)
        fin = np.isfinite(want)
        # the reference sums with numpy's pairwise float64 sum, the oracle in long double
        assert np.allclose(got[fin], want[fin], rtol=1e-12, atol=0), (case, i)
        n_checked += fin.sum()
    assert n_checked > 2000


def test_toi1228_config1_block():
    """BASELINE configs[0]: TOI-1228, TP scenario, N = 1e4, CPU"""
    g = gold("toi1228_ttp.npz")
    got = O.lnl_batch(O.MODEL_TP, g["time"], g["flux"], float(g["sigma"][0]), g["block"])
    assert np.allclose(got, g["out"], rtol=1e-12, atol=0)
    lnL = g["logw"]
    assert abs(O.log_mean_exp(lnL, lnL.size) - g["lnZ"][0]) < 1e-12


# ---------------------------------------------------------------------------------------
# transit model known answers
def test_uniform_disk_closed_forms():
    k = 0.1
    assert abs(O.ma_flux(0.0, k, 0, 0) - (1 - k * k)) < 1e-15
    assert abs(O.ma_flux(0.5, k, 0, 0) - (1 - k * k)) < 1e-15
    assert O.ma_flux(1.1 + 1e-9, k, 0.4, 0.2) == 1.0
    assert O.ma_flux(1.1, k, 0.4, 0.2) == 1.0
    assert O.ma_flux(0.3, 1.5, 0.4, 0.2) == 0.0
    # lens formula on ingress
    z = 1.02
    k0 = np.arccos((k * k + z * z - 1) / (2 * k * z))
    k1 = np.arccos((1 - k * k + z * z) / (2 * z))
    lens = (k * k * k0 + k1 - 0.5 * np.sqrt(4 * z * z - (1 + z * z - k * k) ** 2)) / np.pi
    assert abs(O.ma_flux(z, k, 0, 0) - (1 - lens)) < 1e-15


def test_limb_darkened_centre_closed_form():
    # z = 0: F = 1 - [ (1-c2) k^2 + c2 (2/3)(1 - (1-k^2)^1.5) + u2 k^4/2 ] / (1 - u1/3 - u2/6)
    for k, u1, u2 in ((0.1, 0.4, 0.25), (0.5, 0.6, -0.1), (0.9, 0.2, 0.3)):
        c2 = u1 + 2 * u2
        want = 1 - ((1 - c2) * k * k + c2 * (2 / 3) * (1 - (1 - k * k) ** 1.5) + u2 * k ** 4 / 2) / (1 - u1 / 3 - u2 / 6)
        assert abs(O.ma_flux(0.0, k, u1, u2) - want) < 2e-15


def _quadrature_flux(z, p, u1, u2):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 30
    z, p, u1, u2 = map(mp.mpf, (z, p, u1, u2))
    if z >= 1 + p:
        return mp.mpf(1)

    def intensity(r):
        mu = mp.sqrt(1 - r * r)
        return 1 - u1 * (1 - mu) - u2 * (1 - mu) ** 2

    def half_arc(r):
        if r + z <= p:
            return mp.pi
        if abs(z - r) >= p:
            return mp.mpf(0)
        return mp.acos((r * r + z * z - p * p) / (2 * r * z))

    pts = sorted({max(mp.mpf(0), z - p), abs(p - z), min(mp.mpf(1), z + p)})
    pts = [x for x in pts if 0 <= x <= 1]
    if p > z and pts[0] != 0:
        pts = [mp.mpf(0)] + pts
    occ = mp.quad(lambda r: intensity(r) * 2 * half_arc(r) * r, pts)
    return 1 - occ / (mp.pi * (1 - u1 / 3 - u2 / 6))


def test_flux_against_independent_quadrature():
    rng = np.random.default_rng(7)
    worst = 0.0
    for i in range(60):
        p = 10 ** rng.uniform(-2.5, 0) if i % 3 else rng.uniform(0.02, 0.2)
        mode = i % 5
        if mode == 0:
            z = rng.uniform(0, 1 + p)
        elif mode == 1:
            z = abs(1 - p) + 10 ** rng.uniform(-10, -2) * rng.choice([-1, 1])
        elif mode == 2:
            z = 1 + p - 10 ** rng.uniform(-10, -2)
        elif mode == 3:
            z = p + 10 ** rng.uniform(-12, -2) * rng.choice([-1, 1])
        else:
            z = 10 ** rng.uniform(-10, 0)
        z = abs(z)
        u1, u2 = rng.uniform(0, 0.8), rng.uniform(-0.1, 0.5)
        err = abs(float(O.ma_flux(z, p, u1, u2)) - float(_quadrature_flux(z, p, u1, u2)))
        worst = max(worst, err)
    assert worst < 1e-13, worst
    # exact special points (z = p, z = 1 - p, p = 1/2) take the closed-form branches
    for z, p in ((0.3, 0.3), (0.7, 0.3), (0.5, 0.5), (0.2, 0.8), (0.8, 0.8)):
        err = abs(float(O.ma_flux(z, p, 0.4, 0.25)) - float(_quadrature_flux(z, p, 0.4, 0.25)))
        assert err < 1e-13, (z, p, err)
    # occulter larger than the star (secondary-eclipse regime): coefficients grow like p^4
    for z, p in ((2.5, 3.0), (19.5, 20.0), (20.3, 20.0)):
        err = abs(float(O.ma_flux(z, p, 0.4, 0.25)) - float(_quadrature_flux(z, p, 0.4, 0.25)))
        assert err < 1e-10, (z, p, err)


def test_kepler_and_orbit_conventions():
    rng = np.random.default_rng(3)
    e = rng.uniform(0, 0.95, 500)
    M = rng.uniform(-10, 10, 500)
    E = O.kepler_E(M, e)
    Mr = np.remainder(M + np.pi, 2 * np.pi) - np.pi
    assert np.abs(E - e * np.sin(E) - Mr).max() < 5e-15
    # circular, edge-on: mid-transit at t0, symmetric in time, depth k^2-ish at centre
    t = np.linspace(-0.2, 0.2, 41)
    f = O.evaluate_pv(t, [[0.1, 0.0, 3.0, 10.0, np.pi / 2, 0.0, 0.0]], [[0.0, 0.0]])[0]
    assert abs(f[20] - 0.99) < 1e-15 and np.abs(f - f[::-1]).max() < 1e-14
    # eccentric: w = 90deg - argp puts inferior conjunction at t0 as well
    for w in (0.3, 2.0, 4.5):
        f = O.evaluate_pv(t, [[0.1, 0.0, 3.0, 10.0, np.pi / 2, 0.5, w]], [[0.3, 0.2]])[0]
        assert np.argmin(f) in (19, 20, 21)
    # half a period later the planet is behind the star: no dip
    f = O.evaluate_pv(t + 1.5, [[0.1, 0.0, 3.0, 10.0, np.pi / 2, 0.0, 0.0]], [[0.3, 0.2]])[0]
    assert np.all(f == 1.0)
    # supersampling = mean of sub-exposures at t + exptime*((s-0.5)/S - 0.5)
    S, ex = 7, 0.03
    tt = np.array([0.05])
    sub = np.array([O.evaluate_pv(tt + ex * ((s - 0.5) / S - 0.5), [[0.1, 0.0, 3.0, 10.0, 1.55, 0.2, 1.0]],
                                  [[0.3, 0.2]])[0, 0] for s in range(1, S + 1)])
    got = O.evaluate_pv(tt, [[0.1, 0.0, 3.0, 10.0, 1.55, 0.2, 1.0]], [[0.3, 0.2]], ex, S)[0, 0]
    assert abs(got - sub.mean()) < 1e-15


def test_eccentric_supersampled_row_end_to_end_in_arbitrary_precision():
    """One eccentric, supersampled evaluate_pv row recomputed from first principles in 30-digit
    arithmetic: Kepler's equation by root finding, the sky-projected separation z(t) from the
    orbital elements, the limb-darkened flux by quadrature of the occulted disc, and the mean over
    the S = 20 sub-exposures at t + exptime*((s - 1/2)/S - 1/2).  This pins the CHAIN (not just
    the disc integral) against an independent evaluation.  What it cannot pin are pytransit's own
    conventions, which remain recollections: (1) the orbit convention
    w -> transit at true anomaly f = pi/2 - w with t0 the time of inferior conjunction, (2) the
    far side of the orbit (sin(w + f) < 0) producing no transit, (3) the sub-exposure offsets."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 30
    k, t0, per, a, inc, e, w = 0.11, 0.013, 2.7, 7.5, 1.52, 0.37, 1.1
    u1, u2 = 0.42, 0.23
    S, ex = 20, 0.02
    times = np.array([-0.09, -0.0555, -0.02, 0.013, 0.041, 0.0702, 0.1])

    K, T0, P, A, I, E_, W = map(mp.mpf, (k, t0, per, a, inc, e, w))
    f_tr = mp.pi / 2 - W
    E_tr = 2 * mp.atan(mp.sqrt((1 - E_) / (1 + E_)) * mp.tan(f_tr / 2))
    M_tr = E_tr - E_ * mp.sin(E_tr)

    def z_of(t):
        M = 2 * mp.pi * (mp.mpf(t) - T0) / P + M_tr
        Ecc = mp.findroot(lambda x: x - E_ * mp.sin(x) - M, M)
        f = 2 * mp.atan2(mp.sqrt(1 + E_) * mp.sin(Ecc / 2), mp.sqrt(1 - E_) * mp.cos(Ecc / 2))
        r = A * (1 - E_ ** 2) / (1 + E_ * mp.cos(f))
        s = mp.sin(W + f)
        z = r * mp.sqrt(1 - s ** 2 * mp.sin(I) ** 2)
        return z, s

    want = []
    for t in times:
        acc = mp.mpf(0)
        for s_ in range(1, S + 1):
            ts = mp.mpf(float(t)) + mp.mpf(ex) * ((mp.mpf(s_) - mp.mpf(1) / 2) / S - mp.mpf(1) / 2)
            z, side = z_of(ts)
            acc += mp.mpf(1) if side < 0 else _quadrature_flux(z, K, u1, u2)
        want.append(float(acc / S))
    got = O.evaluate_pv(times, [[k, t0, per, a, inc, e, w]], [[u1, u2]], ex, S)[0]
    assert min(want) < 0.99 and max(want) == 1.0        # the row covers ingress, mid-transit and baseline
    assert np.max(np.abs(got - np.array(want))) < 2e-13, np.abs(got - np.array(want))


def test_depth_bound_of_the_bounded_evaluation_holds_for_the_disc_model():
    """cells_kernel<PRUNE>'s depth screen (csrc/trx_cells.hpp, depth_bound) settles a row from its constants: a body
    of radius ratio k cannot take more than k^2 Imax / Imean of the flux, Imax the largest intensity of the
    quadratic law on the disc, Imean = 1 - u1/3 - u2/6.  The same expression here against the oracle's
    Mandel-Agol flux over all separations, for limb-darkening pairs inside and outside the physical range."""
    rng = np.random.default_rng(5)
    worst = -np.inf
    for _ in range(400):
        k = float(10 ** rng.uniform(-2.5, 0.3))
        u1, u2 = float(rng.uniform(-0.3, 1.0)), float(rng.uniform(-0.3, 0.6))
        if 1 - u1 / 3 - u2 / 6 <= 0.2:
            continue
        imax = max(1.0, 1.0 - u1 - u2)
        if u2 > 0 and u1 < 0:
            imax += u1 * u1 / (4 * u2)
        bound = k * k * imax / (1 - u1 / 3 - u2 / 6)       # (no cap at 1: negative limb intensities can exceed it)
        z = np.concatenate([np.linspace(0, 1 + k, 400), [abs(1 - k), k, 1.0]])
        deficit = max(1.0 - O.ma_flux(float(zz), k, u1, u2) for zz in z if zz < 1 + k)
        worst = max(worst, deficit - bound)
        assert deficit <= bound * (1 + 1e-12) + 1e-15, (k, u1, u2, deficit, bound)
    assert worst <= 1e-15
    # ... and the screen is only consulted below 1 (csrc: depth_screen returns 0 from there on)
