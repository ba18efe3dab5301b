#!/usr/bin/env python3
"""Fixture for the notebook anchors: the numbers the REFERENCE itself printed when it ran with the
real pytransit (stored cell outputs of examples/example.ipynb and examples/kepler_example.ipynb),
next to the inputs those cells used.  They are the only results in the reference tree that passed
through pytransit 2.2's own arithmetic, so they are what
tests/test_gpu_notebook_anchors.py holds the device path against.

Inputs (data files of /root/reference/examples, prepared exactly as the notebook cells do):
  toi465   TOI465_01_lightcurve.csv binned to 100 points (example.ipynb cell 9), P = 3.836169 d,
           the 26-star table of cell 7 (already in toi465_calc_probs.npz), contrast curve of cell 17
  toi411   TOI411_02_lightcurve.csv binned to 100 points (cell 24), P = 4.040051 d, the 32-star
           table of cell 23 (only the target can host the 166 ppm signal)
  kep10    Kepler10b_lightcurve.csv, 477 unbinned points (kepler_example.ipynb cell 9),
           P = 0.837 d, mission = "Kepler", the 6-star table of cell 7
Expected (typed from the stored outputs; they are data):
  per-scenario probabilities of ONE reference run at N = 1e6 (cells 11, 25 / 12), FPP of that run,
  the best-fit planet radius of the TP row, and FPP mean +- std over 20 reference runs for
  TOI-465.01 without (cell 14) and with (cell 18) the contrast curve.

Re-run:  python tests/golden/make_anchors.py   (needs /root/reference/examples)
"""
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
EX = "/root/reference/examples"
nan = np.nan

SCENARIOS = ["TP", "EB", "EBx2P", "PTP", "PEB", "PEBx2P", "STP", "SEB", "SEBx2P", "DTP", "DEB",
             "DEBx2P", "BTP", "BEB", "BEBx2P"]


def binned(fname, n_bins=100):
    """TessLightCurve(time, flux, flux_err).bin(time_bin_size=(tmax - tmin) / 100): fixed-width
    bins from the first stamp, mean flux per bin, error of the mean; sigma = mean binned error"""
    lc = pd.read_csv(os.path.join(EX, fname), header=None)
    t, y, e = (lc[i].values.astype(float) for i in range(3))
    width = (t.max() - t.min()) / n_bins
    idx = np.minimum(((t - t.min()) / width).astype(int), n_bins - 1)
    used = [i for i in range(n_bins) if np.any(idx == i)]
    tb = np.array([t.min() + (i + 0.5) * width for i in used])
    fb = np.array([y[idx == i].mean() for i in used])
    eb = np.array([np.sqrt(np.sum(e[idx == i] ** 2)) / np.sum(idx == i) for i in used])
    return tb, fb, float(np.mean(eb))


def toi411_stars():
    """examples/example.ipynb cell 23 (output of calc_depths(0.000166)); the printed table has no
    J/H/K columns"""
    ids = [100990000, 100990001, 651027929, 100989999, 100989997, 100989996, 100990003, 100992258,
           100990004, 651027983, 100990008, 651027924, 100990007, 100989998, 100990010, 100992262,
           651027992, 651027993, 100992250, 651027921, 651027982, 651027923, 100990011, 651027931,
           651027920, 100989992, 651027482, 100990014, 651027986, 651027919, 100990015, 100989990]
    tmag = [7.7570, 17.5993, 18.7320, 18.1747, 16.6482, 16.0327, 14.3572, 13.2329, 15.9458, 20.7092,
            17.1068, 20.6024, 17.4834, 14.2672, 17.7780, 15.8841, 18.1822, 18.8538, 17.7107, 20.1384,
            20.4606, 19.3605, 18.2129, 20.0574, 20.4112, 16.5542, 18.4864, 14.5485, 18.9344, 20.0979,
            17.4240, 18.0093]
    ra = [54.819841, 54.793531, 54.795230, 54.791855, 54.803113, 54.801156, 54.808848, 54.856880,
          54.838172, 54.833929, 54.827179, 54.826463, 54.794627, 54.774463, 54.797877, 54.861614,
          54.790568, 54.807099, 54.869454, 54.823442, 54.870159, 54.797525, 54.795008, 54.756491,
          54.846882, 54.842338, 54.888120, 54.822924, 54.886834, 54.855863, 54.803639, 54.797750]
    dec = [-42.762551, -42.764160, -42.771759, -42.761727, -42.745878, -42.745691, -42.784554,
           -42.762261, -42.786316, -42.736933, -42.791885, -42.792855, -42.789099, -42.753253,
           -42.794559, -42.742871, -42.731144, -42.725453, -42.783607, -42.805121, -42.740246,
           -42.804805, -42.805048, -42.755694, -42.806245, -42.716863, -42.772374, -42.813591,
           -42.748117, -42.810432, -42.816500, -42.709213]
    mass = [1.17, nan, 0.66, nan, 0.501413, 1.07, nan, nan, 0.66, nan, 1.01, nan, nan, 1.16, 0.309075,
            1.89, nan, nan, 0.399627, nan, nan, nan, 0.59, nan, nan, 1.02, 0.62, 1.06, nan, nan, 0.99, nan]
    rad = [1.116720, nan, 0.537365, nan, 0.503408, 1.198800, 3.494880, 7.990850, 0.596048, nan,
           0.638591, nan, nan, 1.209260, 0.328253, 1.825940, nan, nan, 0.407237, nan, nan, nan,
           0.425768, nan, nan, 0.649991, 0.572985, 1.161500, nan, nan, 0.666959, nan]
    teff = [6161.0, 6245.0, 4205.0, nan, 3449.0, 5910.0, 5040.0, 4760.0, 4218.0, nan, 5668.0, nan, nan,
            6137.0, 3401.0, 7908.0, 5601.0, 3646.0, 3417.0, nan, nan, 4749.0, 3847.0, nan, nan, 5715.0,
            3992.0, 5870.0, 4030.0, nan, 5606.0, 3883.0]
    plx = [15.899900, 0.492555, -0.107290, nan, nan, 0.298758, 0.319435, 0.260391, 1.477730, nan,
           0.351775, nan, nan, 0.715483, 1.974120, 0.001260, -0.163846, 2.157550, 1.362360, 0.448897,
           nan, 0.307366, 0.867494, 2.231890, -0.201713, 0.500855, 0.262623, 0.705178, 0.762538,
           0.802713, 0.194394, 1.127660]
    fr = [9.999181e-01, 7.972274e-06, 8.916198e-07, 3.097199e-06, 1.950503e-05, 2.098939e-05,
          1.293266e-05, 1.378523e-05, 2.638980e-06, 1.717102e-08, 5.900667e-08, 1.071358e-09,
          1.378635e-10, 1.837653e-09, 3.512156e-12, 9.318788e-11, 1.893848e-11, 7.346003e-12,
          6.457540e-12, 1.325042e-15, 1.010465e-16, 1.183616e-18, 6.155347e-19, 3.622098e-19,
          7.213192e-19, 2.632556e-17, 1.112206e-18, 1.800458e-19, 1.345936e-22, 5.313263e-24,
          1.544053e-25, 1.367531e-23]
    n = len(ids)
    return pd.DataFrame({"ID": ids, "Tmag": tmag, "Jmag": [nan] * n, "Hmag": [nan] * n, "Kmag": [nan] * n,
                         "ra": ra, "dec": dec, "mass": mass, "rad": rad, "Teff": teff, "plx": plx,
                         "fluxratio": fr, "tdepth": [0.000166] + [0.0] * (n - 1)})


def kep10_stars():
    """examples/kepler_example.ipynb cell 7 (output of calc_depths(0.00019))"""
    return pd.DataFrame({
        "ID": [377780790, 1717218059, 1717218056, 1717218060, 377780779, 1717218057],
        "Tmag": [10.4767, 17.8806, 20.0671, 17.4027, 15.8564, 18.4788],
        "Jmag": [9.889, nan, nan, nan, 14.727, nan], "Hmag": [9.563, nan, nan, nan, 14.117, nan],
        "Kmag": [9.496, nan, nan, nan, 14.075, nan],
        "ra": [285.679422, 285.680619, 285.677382, 285.680220, 285.685892, 285.682207],
        "dec": [50.241306, 50.245790, 50.248546, 50.249945, 50.249906, 50.251926],
        "mass": [1.017, 1.070, nan, 1.030, 0.700, nan],
        "rad": [1.089740, 0.809877, nan, 1.055070, 0.804521, nan],
        "Teff": [5706.0, 5895.0, nan, 5771.0, 4467.0, 4923.0],
        "plx": [5.361850, -0.111711, 0.879011, -0.004017, 0.999995, 0.325102],
        "fluxratio": [9.999993e-01, 6.626514e-07, 1.308192e-14, 4.187665e-19, 2.692696e-22, 4.131923e-30],
        "tdepth": [0.00019, 0.0, 0.0, 0.0, 0.0, 0.0]})


# one reference run each, N = 1e6 (probabilities in the order of SCENARIOS)
PROB_465 = [8.969353e-01, 0.0, 2.788198e-42, 7.212812e-02, 8.221751e-281, 1.607920e-37, 2.565628e-02,
            3.151474e-14, 1.641204e-44, 5.280339e-03, 0.0, 4.519202e-48, 4.979879e-13, 2.463958e-31,
            3.805488e-37]                                                   # example.ipynb cell 11
PROB_411 = [7.506777e-01, 1.139451e-58, 1.811696e-48, 1.188953e-01, 1.765332e-60, 2.087916e-53,
            3.378508e-02, 5.009660e-18, 4.586207e-15, 9.049978e-02, 3.327247e-61, 3.731821e-52,
            1.001123e-04, 5.996511e-03, 4.546522e-05]                       # example.ipynb cell 25
PROB_K10 = [9.986258e-01, 0.0, 0.0, 1.322033e-03, 0.0, 0.0, 8.359147e-06, 0.0, 0.0, 4.379518e-05, 0.0,
            0.0, 3.686393e-152, 0.0, 0.0]                                   # kepler_example.ipynb cell 12


def main():
    out = {"scenarios": np.array(SCENARIOS)}
    # TOI-465.01: light curve, stars and contrast curve are in toi465_calc_probs.npz / toi465_cc.csv
    out["toi465_prob"] = np.array(PROB_465)
    out["toi465_FPP"] = np.array([0.0257])
    out["toi465_Rp_TP"] = np.array([6.247005])
    out["toi465_FPP20_nocc"] = np.array([0.0432, 0.0578])        # cell 14: mean, std of 20 runs
    out["toi465_FPP20_cc"] = np.array([0.0032, 0.005])           # cell 18
    tb, fb, sg = binned("TOI411_02_lightcurve.csv")
    out.update({"toi411_time": tb, "toi411_flux": fb, "toi411_sigma": np.array([sg]),
                "toi411_P_orb": np.array([4.040051]), "toi411_prob": np.array(PROB_411),
                "toi411_FPP": np.array([0.0399]), "toi411_Rp_TP": np.array([1.606318])})
    st = toi411_stars()
    for c in st.columns:
        out["toi411_stars_" + c] = st[c].values.astype(float)
    lc = pd.read_csv(os.path.join(EX, "Kepler10b_lightcurve.csv"), header=None)
    t, y, e = (lc[i].values.astype(float) for i in range(3))
    keep = ~np.isnan(y)                                          # kepler_example.ipynb cell 9
    out.update({"kep10_time": t[keep], "kep10_flux": y[keep], "kep10_sigma": np.array([np.mean(e[keep])]),
                "kep10_P_orb": np.array([0.837]), "kep10_prob": np.array(PROB_K10),
                "kep10_FPP": np.array([8.359147213754525e-06]), "kep10_Rp_TP": np.array([1.530357])})
    st = kep10_stars()
    for c in st.columns:
        out["kep10_stars_" + c] = st[c].values.astype(float)
    np.savez_compressed(os.path.join(HERE, "notebook_anchors.npz"), **out)
    # the two unbinned example light curves (data files of the reference's examples/): the input of the
    # binning variants of profiles/anchor_sensitivity.py (columns: time, flux, flux_err)
    raw = {}
    for key, fname in (("toi411", "TOI411_02_lightcurve.csv"), ("toi465", "TOI465_01_lightcurve.csv")):
        raw[key] = pd.read_csv(os.path.join(EX, fname), header=None).values.astype(float)
    np.savez_compressed(os.path.join(HERE, "example_lightcurves.npz"), **raw)
    print("toi411: %d points, sigma %.3e; kep10: %d points, sigma %.3e"
          % (tb.size, sg, keep.sum(), out["kep10_sigma"][0]))


if __name__ == "__main__":
    main()
