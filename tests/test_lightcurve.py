"""Light-curve preparation (triceratops_amd/lightcurve.py): the fold / trim / bin step the
reference's notebooks delegate to lightkurve, checked by its defining properties and against a
direct per-bin loop."""
import numpy as np
import pytest

from triceratops_amd import lightcurve as lc


def test_fold_wraps_to_half_a_period_around_the_midpoint():
    t = np.array([100.0, 103.3, 106.6, 101.0, 104.9, 99.9])
    ph, order = lc.fold(t, 3.3, 100.0)
    assert np.all(np.diff(ph) >= 0) and np.all((ph >= -1.65) & (ph < 1.65))
    assert np.allclose(ph, [-0.1, 0.0, 0.0, 0.0, 1.0, 1.6], atol=1e-9)
    got = t[order]
    assert got[0] == 99.9 and sorted(got[1:4]) == [100.0, 103.3, 106.6] and list(got[4:]) == [101.0, 104.9]


def test_bins_are_fixed_width_means_with_nan_for_empty_bins():
    rng = np.random.default_rng(0)
    t = np.sort(np.concatenate([rng.uniform(-0.4, -0.1, 300), rng.uniform(0.05, 0.4, 300)]))
    y = 1 + 1e-3 * rng.standard_normal(t.size)
    w = 0.01
    c, m, n = lc.bin_lightcurve(t, y, time_bin_size=w)
    assert np.allclose(np.diff(c), w) and abs(c[0] - (t.min() + w / 2)) < 1e-15
    assert n.sum() == t.size
    for i in range(c.size):
        sel = (t >= t.min() + i * w) & (t < t.min() + (i + 1) * w) if i < c.size - 1 else t >= t.min() + i * w
        assert n[i] == sel.sum()
        assert (np.isnan(m[i]) and not sel.any()) or abs(m[i] - y[sel].mean()) < 1e-14
    assert np.isnan(m).sum() >= 10                       # the gap between -0.1 and 0.05
    assert abs(np.nansum(m * n) - y.sum()) < 1e-9         # binning preserves the total
    c2, m2, n2 = lc.bin_lightcurve(t, y, n_bins=37)
    assert c2.size == 37 and n2.sum() == t.size
    y[5] = np.nan
    assert lc.bin_lightcurve(t, y, time_bin_size=w)[2].sum() == t.size - 1


def test_argument_and_edge_cases():
    assert lc.bin_lightcurve([], [], n_bins=3)[0].size == 0
    with pytest.raises(ValueError):
        lc.bin_lightcurve([0.0, 1.0], [1.0, 1.0])
    with pytest.raises(ValueError):
        lc.bin_lightcurve([0.0, 1.0], [1.0, 1.0], time_bin_size=-1.0)
    c, m, n = lc.bin_lightcurve([2.0, 2.0], [1.0, 3.0], n_bins=4)      # zero time span
    assert n[0] == 2 and m[0] == 2.0
    with pytest.raises(ValueError):
        lc.prepare([1.0, 2.0], [1.0, 1.0], half_width=0.4)


def test_prepare_gives_calc_probs_inputs():
    rng = np.random.default_rng(1)
    t = rng.uniform(-1.5, 1.5, 20000)
    y = 1 + 4e-4 * rng.standard_normal(t.size) - 5e-4 * (np.abs(t) < 0.08)
    tb, yb, sg = lc.prepare(t, y, half_width=0.4, n_bins=200)
    assert 195 <= tb.size <= 201 and not np.isnan(yb).any() and np.all(np.abs(tb) < 0.41)
    assert 0.5 < sg / (4e-4 / np.sqrt(20000 * 0.8 / 3 / 200)) < 2.0
    assert yb[np.abs(tb) < 0.05].mean() < yb[np.abs(tb) > 0.2].mean() - 3e-4
