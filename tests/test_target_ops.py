"""The best-fit curves that plot_fits draws (triceratops.py:1487-1638) against the
line data of the reference's own figure (make_golden.py section 7), and calc_depths' pixel integrals.
CPU variant on the oracle stand-in, GPU variant on the kernels."""
import os

import numpy as np
import pandas as pd
import pytest

from helpers import GOLD, gold, install_cpu_device_fakes

G = gold("target_ops.npz")
C = gold("calc_probs.npz")


def _target():
    from triceratops_amd.triceratops import target
    stars = pd.DataFrame({
        "ID": ["111", "222", "333", "444"], "Tmag": [10.4, 13.0, 15.5, 12.2],
        "Jmag": [9.5, 12.1, 14.6, 11.5], "Hmag": [9.1, 11.7, 14.2, 11.1],
        "Kmag": [9.0, 11.6, 14.1, 11.0], "ra": [10.0, 10.004, 10.01, 9.99],
        "dec": [-5.0, -5.003, -5.01, -4.995], "mass": [0.82, 0.6, np.nan, 1.1],
        "rad": [0.8, 0.58, np.nan, 1.3], "Teff": [5100.0, 4000.0, np.nan, 6000.0],
        "plx": [14.2, 3.0, np.nan, 2.0]})
    pix = [np.array([[10.2, 10.7], [11.9, 11.3], [14.0, 7.5], [8.4, 12.6]]),
           np.array([[10.6, 10.1], [12.2, 10.9], [14.5, 7.0], [8.9, 12.0]])]
    tg = target(111, np.array([1]), stars=stars, pix_coords=pix,
                trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"))
    tg.calc_depths(0.007, [C["aperture0"], C["aperture1"]])
    return tg


def _same(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def _fit_case(tol):
    tg = _target()
    np.random.seed(int(G["fit_seed"][0]))
    tg.calc_probs(C["time"], C["flux"], float(C["sigma"][0]), 3.3, N=1500, parallel=True, verbose=0,
                  contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"),
                  drop_scenario=["PEB"])
    mt, curves = tg.fit_curves(C["time"], C["flux"], float(C["sigma"][0]))
    assert np.array_equal(mt, G["fit_model_time"]) and len(curves) == len(G["fit_model"])
    for c, want, data, labels in zip(curves, G["fit_model"], G["fit_data"], G["fit_labels"]):
        assert [str(c["ID"]), str(c["scenario"])] == list(labels)
        assert np.allclose(c["flux"], data, rtol=1e-14, atol=0)
        assert np.max(np.abs(c["model"] - want)) < tol, c["scenario"]
    assert np.all(curves[4]["model"] == 1.0) and np.all(curves[5]["model"] == 1.0)   # dropped PEB
    assert any(np.min(c["model"]) < 0.999 for c in curves)
    return tg


def test_fit_curves_match_reference_plot_on_host_fakes(monkeypatch):
    install_cpu_device_fakes(monkeypatch)
    _fit_case(1e-12)


@pytest.mark.gpu
def test_fit_curves_match_reference_plot_on_gpu():
    _fit_case(1e-9)


def test_calc_depths_pixel_integrals_equal_numerical_integration_of_gauss2d():
    """reference tests/test_analytic_psf.py: the closed-form aperture integral of calc_depths
    equals scipy's dblquad of the Gaussian profile over the same pixels (to dblquad's accuracy)"""
    from scipy.integrate import dblquad

    def gauss2d(x, y, mu_x, mu_y, sigma, A):         # the PSF profile of the reference's funcs.Gauss2D
        return A / (2 * np.pi * sigma ** 2) * np.exp(-((x - mu_x) ** 2 + (y - mu_y) ** 2) / (2 * sigma ** 2))

    tg = _target()
    ap = C["aperture0"]
    tg.pix_coords, tg.sectors = tg.pix_coords[:1], np.array([1])
    tg.calc_depths(0.007, [ap])
    Tmag = tg.stars["Tmag"].values
    amp = 10 ** ((Tmag.min() - Tmag) / 2.5)
    rel = np.array([sum(dblquad(gauss2d, y - 0.5, y + 0.5, x - 0.5, x + 0.5,
                                args=(mx, my, 0.75, A))[0] for x, y in ap)
                    for (mx, my), A in zip(tg.pix_coords[0], amp)])
    assert np.allclose(tg.stars["fluxratio"].values, rel / rel.sum(), rtol=0, atol=1e-8)
    assert abs(tg.stars["fluxratio"].sum() - 1) < 1e-14


def test_prepare_lists_the_same_units_for_a_star_table_with_a_text_column():
    """target._prepare takes the star table's numeric columns in ONE float64 conversion (eleven pandas column reads
    were 30 us of every target's preparation, on every rank of a batch); a table that holds a non-numeric column --
    string IDs, a user's notes -- cannot be converted whole and takes its columns one by one as before.  Both ways
    must list the same units: same rows, names, star numbers, IDs, weights, (job, star) tags, the same table size
    (triceratops.py:673-735: the star filter and table sizes of calc_probs)."""
    from triceratops_amd.triceratops import target
    stars = pd.DataFrame({
        "ID": [101, 102, 103], "Tmag": [10.0, 13.0, 14.0], "Jmag": [9.2, 12.1, 13.0], "Hmag": [8.8, 11.6, 12.5],
        "Kmag": [8.7, 11.5, 12.4], "ra": [10.0, 10.004, 10.01], "dec": [-5.0, -5.003, -5.01],
        "mass": [1.0, 0.6, np.nan], "rad": [1.0, 0.6, np.nan], "Teff": [5700, 4000, 3900], "plx": [10.0, 2.0, 1.0],
        "fluxratio": [0.9, 0.07, 0.03], "tdepth": [0.0011, 0.014, 0.0]})
    t = np.linspace(-0.2, 0.2, 60)
    f = 1.0 - 0.001 * (np.abs(t) < 0.05)
    f[7] = np.nan                                                  # (the reference's NaN filter, :673-676)
    kw = dict(time=t, flux_0=f, flux_err_0=5e-4, P_orb=3.0, N=1000, job=3)
    tri = os.path.join(GOLD, "trilegal_synth.csv")
    a = target(101, np.array([1]), stars=stars.copy(), trilegal_fname=tri)
    with_text = stars.copy()
    with_text["note"] = ["target", "neighbour", "faint"]
    b = target(101, np.array([1]), stars=with_text, trilegal_fname=tri)
    ua, na = a._prepare(**kw)
    ub, nb = b._prepare(**kw)
    assert na == nb == 3 * 2 + 12 and len(ua) == len(ub) == 10 + 2
    for x, y in zip(ua, ub):
        assert x[:4] == y[:4] and x[5:] == y[5:] and (x[4] is None) == (y[4] is None)
    assert [u[8] for u in ua] == [(3, 0)] * 10 + [(3, 1), (3, 1)]
