"""Largest differences between two directories of arrays written by ab_bits.py (python ab_bits.py <rows> <dir>)."""
import glob
import os
import sys

import numpy as np

a, b = sys.argv[1:3]
worst_h = worst_g = 0.0
for f in sorted(glob.glob(os.path.join(a, "*.npz"))):
    x, y = np.load(f), np.load(os.path.join(b, os.path.basename(f)))
    fin = np.isfinite(x["h"])
    assert np.array_equal(fin, np.isfinite(y["h"])), f
    dh = float(np.max(np.abs(x["h"][fin] - y["h"][fin]) / np.abs(x["h"][fin]))) if fin.any() else 0.0
    dg = float(np.nanmax(np.abs(x["g"] - y["g"])))
    worst_h, worst_g = max(worst_h, dh), max(worst_g, dg)
    print("%-28s chi2/2 rel %.2e   flux abs %.2e   (rows differing %d of %d)" % (
        os.path.basename(f), dh, dg, int((x["h"][fin] != y["h"][fin]).sum()), int(fin.sum())))
print("worst: chi2/2 rel %.2e, flux abs %.2e" % (worst_h, worst_g))
