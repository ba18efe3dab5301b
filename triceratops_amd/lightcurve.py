"""Light-curve preparation in front of calc_probs: fold, trim and bin.

The reference has no code for this step: its notebooks hand it to lightkurve
(examples/TSCIII_tutorial.ipynb cell 5: `TessLightCurve(time, flux).bin(time_bin_size=...)` on the
points with |t| < 0.4 d; examples/example.ipynb folds with `lc.fold`).  lightkurve is not a
dependency here, so the same three operations are provided on plain numpy arrays; calc_probs
takes their output directly (it drops the NaN of empty bins itself, triceratops.py:707-709).
Bin edges follow lightkurve 2.x's fixed-width binning [recollection -- lightkurve is not in this
image]: bins of `time_bin_size` start at the first time stamp, each bin reports its centre and the
nan-mean of its fluxes, an empty bin reports NaN.
"""
import numpy as np


def fold(time, period: float, epoch_time: float):
    """Days from the nearest transit midpoint, in [-period/2, period/2), sorted; returns
    (folded_time, order) so that flux[order] lines up with folded_time."""
    time = np.asarray(time, dtype=np.float64)
    phase = np.mod(time - epoch_time + 0.5 * period, period) - 0.5 * period
    order = np.argsort(phase, kind="stable")
    return phase[order], order


def trim(time, flux, half_width: float):
    """The points within `half_width` days of the transit midpoint."""
    time, flux = np.asarray(time, dtype=np.float64), np.asarray(flux, dtype=np.float64)
    keep = np.abs(time) < half_width
    return time[keep], flux[keep]


def bin_lightcurve(time, flux, time_bin_size: float = None, n_bins: int = None):
    """Fixed-width bins starting at min(time).  Give the width, or a bin count (the width is then
    the time span / n_bins).  Returns (bin centres, mean flux per bin, points per bin); empty bins
    hold NaN."""
    time, flux = np.asarray(time, dtype=np.float64), np.asarray(flux, dtype=np.float64)
    if time.size == 0:
        return np.empty(0), np.empty(0), np.empty(0, dtype=np.int64)
    if (time_bin_size is None) == (n_bins is None):
        raise ValueError("give exactly one of time_bin_size and n_bins")
    t0, span = np.min(time), np.max(time) - np.min(time)
    if time_bin_size is None:
        time_bin_size = span / n_bins if span > 0 else 1.0
    if not time_bin_size > 0:
        raise ValueError("time_bin_size must be positive")
    if n_bins is None:
        n_bins = int(np.floor(span / time_bin_size)) + 1
    idx = np.minimum(((time - t0) / time_bin_size).astype(np.int64), n_bins - 1)
    ok = ~np.isnan(flux)
    count = np.bincount(idx[ok], minlength=n_bins)
    total = np.bincount(idx[ok], weights=flux[ok], minlength=n_bins)
    mean = np.full(n_bins, np.nan)
    np.divide(total, count, out=mean, where=count > 0)
    centres = t0 + (np.arange(n_bins) + 0.5) * time_bin_size
    return centres, mean, count


def prepare(time, flux, half_width: float = 0.4, n_bins: int = 200, n_sigma: int = 50):
    """The tutorial's preparation in one call: trim to |t| < half_width, bin to `n_bins` bins of
    width 2 max|t| / n_bins, drop empty bins, and estimate the per-point uncertainty as the
    standard deviation of the first `n_sigma` (out-of-transit) binned points.
    Returns (time, flux, sigma) ready for calc_probs."""
    t, y = trim(time, flux, half_width)
    if t.size == 0:
        raise ValueError("no points within %g d of the transit midpoint" % half_width)
    tb, yb, _ = bin_lightcurve(t, y, time_bin_size=2 * np.max(t) / n_bins)
    keep = ~np.isnan(yb)
    tb, yb = tb[keep], yb[keep]
    return tb, yb, float(np.std(yb[:n_sigma]))
