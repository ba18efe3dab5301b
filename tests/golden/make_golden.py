#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python
(/root/reference/triceratops: priors.py, funcs.py, likelihoods.py, marginal_likelihoods.py,
_numerics.py) in the build container.

The reference cannot be imported as-is here (astropy, pytransit, numba, mechanicalsoup, bs4 are
not installed and there is no network), so this script installs throw-away `sys.modules` shims
-- five astropy cgs constants, empty stubs for the web/FITS modules, and the CPU oracle's
`QuadraticModel` at the pytransit seam -- and then runs the reference unmodified.  Only DATA
leaves this script: inputs, seeds, every per-draw block the reference hands to its likelihood
functions, their outputs, the log-weights it reduces, lnZ and the best-fit tables.  No reference
source is copied.  Re-run:  python tests/golden/make_golden.py   (needs /root/reference).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import oracle as O  # noqa: E402


def install_shims():
    class _Q:
        def __init__(self, v):
            self.cgs = types.SimpleNamespace(value=v)

    const = types.ModuleType("astropy.constants")
    const.G = _Q(6.6743e-08)
    const.M_sun = _Q(1.988409870698051e+33)
    const.R_sun = _Q(69570000000.0)
    const.R_earth = _Q(637810000.0)
    const.au = _Q(14959787070000.0)
    astropy = types.ModuleType("astropy")
    astropy.constants = const
    io = types.ModuleType("astropy.io")
    fits = types.ModuleType("astropy.io.fits")
    io.fits = fits
    astropy.io = io
    ms = types.ModuleType("mechanicalsoup")
    ms.StatefulBrowser = object
    bs4 = types.ModuleType("bs4")
    bs4.BeautifulSoup = object
    pt = types.ModuleType("pytransit")
    pt.QuadraticModel = O.QuadraticModel
    sys.modules.update({"astropy": astropy, "astropy.constants": const, "astropy.io": io,
                        "astropy.io.fits": fits, "mechanicalsoup": ms, "bs4": bs4, "pytransit": pt})
    sys.path.insert(0, REF)


def write_trilegal(path, rng, n=400):
    """synthetic TRILEGAL table with the columns funcs.trilegal_results reads + 2 trailer rows"""
    import pandas as pd
    mass = rng.uniform(0.12, 1.6, n)
    logg = rng.uniform(3.2, 5.1, n)
    logTe = np.log10(rng.uniform(3000, 11000, n))
    mh = rng.uniform(-1.5, 0.4, n)
    tess = rng.uniform(9.0, 21.0, n)
    j = tess - rng.uniform(0.3, 1.2, n)
    h = j - rng.uniform(0.1, 0.7, n)
    ks = h - rng.uniform(0.0, 0.3, n)
    df = pd.DataFrame({"Mact": mass, "logg": logg, "logTe": logTe, "[M/H]": mh, "TESS": tess,
                       "J": j, "H": h, "Ks": ks})
    trailer = pd.DataFrame({c: [np.nan, np.nan] for c in df.columns})
    pd.concat([df, trailer]).to_csv(path)


def write_contrast_curve(path):
    sep = np.linspace(0.05, 4.0, 40)
    con = 8.0 * (1 - np.exp(-sep / 0.8))
    np.savetxt(path, np.stack([sep, con]).T, delimiter=",", fmt="%.6f")


def synthetic_light_curve(n_time, seed):
    rng = np.random.default_rng(seed)
    t = np.linspace(-0.2, 0.2, n_time)
    Rsun, Rearth = 69570000000.0, 637810000.0
    row = np.array([[0.085 * Rsun / Rearth], [3.3], [88.0], [11.0 * 0.8 * Rsun], [0.8], [0.45],
                    [0.2], [0.0], [0.0], [0.0]])
    curve = O.flux_grid(O.MODEL_TP, t, row)[0][0]
    sigma = 6e-4
    return t, curve + rng.normal(0, sigma, n_time), sigma


def binned_toi1228(n_bins=200, half_width=0.4):
    """TOI-1228 folded light curve of the reference's examples/, prepared like
    examples/TSCIII_tutorial.ipynb cells 3-5 and 20: flux = y + 1, |t| < 0.4 d, 200 equal time
    bins (mean per bin; the notebook uses lightkurve's .bin), sigma = std of the first 50 bins."""
    import pandas as pd
    lc = pd.read_csv(os.path.join(REF, "examples", "TOI1228_folded_lightcurve.csv"))
    t, y = lc.x_fold.values.astype(float), lc.y.values.astype(float) + 1.0
    keep = np.abs(t) < half_width
    t, y = t[keep], y[keep]
    width = 2 * np.max(t) / n_bins
    idx = np.minimum(((t - t.min()) / width).astype(int), n_bins - 1)
    tb = np.array([t[idx == i].mean() for i in range(n_bins) if np.any(idx == i)])
    fb = np.array([y[idx == i].mean() for i in range(n_bins) if np.any(idx == i)])
    return tb, fb, float(np.std(fb[:50]))


def import_reference_target():
    """triceratops/triceratops.py imports the catalogue / plotting stack at module level; stub it.
    Objects are created without running __init__ (which queries MAST/TessCut/TRILEGAL)."""
    for name in ("lightkurve", "astroquery", "astroquery.mast", "astroquery.vizier",
                 "astropy.coordinates", "astropy.wcs", "astropy.wcs.utils", "astropy.units"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["astroquery.mast"].Catalogs = object
    sys.modules["astroquery.mast"].Tesscut = object
    sys.modules["astroquery.vizier"].Vizier = object
    sys.modules["astropy.coordinates"].SkyCoord = object
    sys.modules["astropy.wcs"].WCS = object
    sys.modules["astropy.wcs.utils"].pixel_to_skycoord = object
    import matplotlib
    matplotlib.use("Agg")
    import triceratops.triceratops as rtr
    return rtr


def main():
    install_shims()
    if "--only-toi465" in sys.argv:
        toi465(import_reference_target(), os.path.join(HERE, "trilegal_synth.csv"))
        return
    import triceratops.funcs as rfuncs
    import triceratops.likelihoods as rlik
    import triceratops.marginal_likelihoods as rml
    import triceratops.priors as rpri
    from triceratops import _numerics as rnum

    out = {}
    rng = np.random.default_rng(1234)

    # ---- (1) priors / funcs on fixed inputs ------------------------------------------------
    x = np.concatenate([rng.uniform(0, 1, 500), [0.0, 1.0, 1e-12, 1 - 1e-12]])
    out["uniforms"] = x
    for M in (1.3, 1.0, 0.7, 0.3, 0.25, 0.1, 0.05):
        out["sample_q_%g" % M] = rpri.sample_q(x.copy(), M)
        out["sample_qc_%g" % M] = rpri.sample_q_companion(x.copy(), M)
    Ms = rng.uniform(0.1, 1.5, x.size)
    out["rp_masses"] = Ms
    out["sample_rp"] = rpri.sample_rp(x.copy(), Ms, False)
    out["sample_rp_flat"] = rpri.sample_rp(x.copy(), Ms, True)
    out["sample_inc"] = rpri.sample_inc(x.copy())
    out["sample_w"] = rpri.sample_w(x.copy())
    for planet, P in ((True, 3.0), (False, 3.0), (False, 20.0)):
        np.random.seed(77)
        out["sample_ecc_%d_%g" % (planet, P)] = rpri.sample_ecc(x.copy(), planet, P)
    masses = np.concatenate([rng.uniform(0.05, 3.0, 300), [0.63, 0.1, 0.26]])
    out["sr_masses"] = masses
    r_, t_ = rfuncs.stellar_relations(masses, np.full(masses.size, 1.1), np.full(masses.size, 6100.0))
    out["sr_radii"], out["sr_teffs"] = r_, t_
    for band in ("TESS", "Vis", "J", "H", "K"):
        out["flux_relation_" + band] = rfuncs.flux_relation(masses, band)
    dm = np.abs(rng.normal(3, 2.5, 400))
    out["prior_dmags"] = dm
    cc_path = os.path.join(HERE, "contrast_curve_synth.csv")
    write_contrast_curve(cc_path)
    seps, cons = rfuncs.file_to_contrast_curve(cc_path)
    for M in (1.25, 0.8):
        for plx in (12.5, np.nan):
            tag = "%g_%s" % (M, "nan" if np.isnan(plx) else "%g" % plx)
            out["bound_TP_" + tag] = rpri.lnprior_bound_TP(M, plx, dm, seps, cons)
            out["bound_EB_" + tag] = rpri.lnprior_bound_EB(M, plx, dm, seps, cons)
            out["bound_TP_nocc_" + tag] = rpri.lnprior_bound_TP(M, plx, dm, np.array([2.2]), np.array([1.0]))
    out["background"] = rpri.lnprior_background(391, dm, seps, cons)
    fl, fe = rfuncs.renorm_flux(np.array([1.0, 0.999, 0.9985]), 5e-4, 0.37)
    out["renorm_flux"], out["renorm_err"] = fl, np.array([fe])
    np.savez_compressed(os.path.join(HERE, "priors_funcs.npz"), **out)

    # ---- (2) _numerics -------------------------------------------------------------------------
    num = {}
    vecs = {
        "a": np.full(1000, -2000.0),
        "b": np.array([-1001.0, -1002.0, -1003.0, -1004.0, -1005.0] + [-np.inf] * 5),
        "c": np.array([-1.0] + [-np.inf] * 9),
        "d": np.full(10, -np.inf),
        "e": np.array([-1.0, np.nan, -np.inf, np.nan, -np.inf]),
        "f": np.array([-1.0, np.inf, -3.0]),
        "g": rng.uniform(-3000, -1, 5000),
    }
    vecs["g"][rng.uniform(size=5000) < 0.9] = -np.inf
    for k, v in vecs.items():
        num["lme_in_" + k] = v
        num["lme_out_" + k] = np.array([rnum._log_mean_exp(v, N_total=v.size)])
    lz = {"ok": np.array([-10.0, -11.0, -np.inf, -9.5]), "allneg": np.full(4, -np.inf),
          "anom": np.array([-1.0, np.nan, -2.0])}
    for k, v in lz.items():
        p, st = rnum._normalize_probabilities(v)
        num["norm_in_" + k], num["norm_out_" + k], num["norm_status_" + k] = v, p, np.array([st])
    np.savez_compressed(os.path.join(HERE, "numerics.npz"), **num)

    # ---- (3) the ten lnZ_* with every kernel-side block captured -----------------------------
    tri_path = os.path.join(HERE, "trilegal_synth.csv")
    write_trilegal(tri_path, np.random.default_rng(99))
    t, f, sigma = synthetic_light_curve(60, 5)
    star = dict(M_s=0.82, R_s=0.8, Teff=5100.0, Z=0.0, plx=14.2, Tmag=10.4, Jmag=9.5, Hmag=9.1,
                Kmag=9.0)
    captured = []

    def wrap(name):
        fn = getattr(rlik, name)

        def inner(time, flux, sig, *cols, **kw):
            args = [np.array(c, dtype=float, copy=True) for c in cols]
            res = fn(time, flux, sig, *cols, **kw)
            captured.append((name, args, dict(kw), np.array(res, copy=True)))
            return res
        return inner

    logws = []

    def lme(logw, *, N_total):
        logws.append(np.array(logw, copy=True))
        return rnum._log_mean_exp(logw, N_total=N_total)

    for nm in ("lnL_TP_p", "lnL_EB_p", "lnL_EB_twin_p", "lnL_TP", "lnL_EB", "lnL_EB_twin"):
        setattr(rml, nm, wrap(nm))
    rml._log_mean_exp = lme

    N = 2000
    b = (t, f, sigma)
    s = star
    plan = {
        "TTP": lambda P, par: rml.lnZ_TTP(*b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], N, par),
        "TEB": lambda P, par: rml.lnZ_TEB(*b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], N, par),
        "PTP": lambda P, par, cc=None, filt="TESS": rml.lnZ_PTP(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["plx"], cc, filt, N, par),
        "PEB": lambda P, par, cc=None, filt="TESS": rml.lnZ_PEB(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["plx"], cc, filt, N, par),
        "STP": lambda P, par, cc=None, filt="TESS": rml.lnZ_STP(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["plx"], cc, filt, N, par),
        "SEB": lambda P, par, cc=None, filt="TESS": rml.lnZ_SEB(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["plx"], cc, filt, N, par),
        "DTP": lambda P, par, cc=None, filt="TESS": rml.lnZ_DTP(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["Tmag"], s["Jmag"], s["Hmag"],
            s["Kmag"], tri_path, cc, filt, N, par),
        "DEB": lambda P, par, cc=None, filt="TESS": rml.lnZ_DEB(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Z"], s["Tmag"], s["Jmag"], s["Hmag"],
            s["Kmag"], tri_path, cc, filt, N, par),
        "BTP": lambda P, par, cc=None, filt="TESS": rml.lnZ_BTP(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Tmag"], s["Jmag"], s["Hmag"], s["Kmag"],
            tri_path, cc, filt, N, par),
        "BEB": lambda P, par, cc=None, filt="TESS": rml.lnZ_BEB(
            *b, P, s["M_s"], s["R_s"], s["Teff"], s["Tmag"], s["Jmag"], s["Hmag"], s["Kmag"],
            tri_path, cc, filt, N, par),
    }
    gold = {"time": t, "flux": f, "sigma": np.array([sigma]), "N": np.array([N]),
            "star": np.array([star[k] for k in ("M_s", "R_s", "Teff", "Z", "plx", "Tmag", "Jmag",
                                                "Hmag", "Kmag")])}
    cases = []
    seed = 1000
    for name, fn in plan.items():
        variants = [("par", 3.3, True, {}), ("range", [2.5, 4.0], True, {})]
        if name not in ("TTP", "TEB"):
            variants.append(("ccJ", 3.3, True, {"cc": cc_path, "filt": "J"}))
        variants.append(("serial", 3.3, False, {}))
        for vname, P, par, kw in variants:
            seed += 1
            np.random.seed(seed)
            del captured[:], logws[:]
            if par:
                res = fn(P, par, **kw)
            else:
                # the serial path is a Python loop over draws in the reference: smaller N
                res = _with_N(plan, name, rml, b, s, tri_path, 300, P, kw)
            case = "%s_%s" % (name, vname)
            cases.append(case)
            dicts = res if isinstance(res, tuple) else (res,)
            gold[case + "_seed"] = np.array([seed])
            gold[case + "_nres"] = np.array([len(dicts)])
            for i, d in enumerate(dicts):
                gold["%s_lnZ%d" % (case, i)] = np.array([d["lnZ"]])
                for k, v in d.items():
                    if k != "lnZ":
                        gold["%s_res%d_%s" % (case, i, k)] = np.asarray(v, dtype=float)
            for i, lw in enumerate(logws):
                gold["%s_logw%d" % (case, i)] = lw
            if par:
                for i, (cname, args, kw_, outv) in enumerate(captured):
                    gold["%s_call%d_name" % (case, i)] = np.array([cname])
                    gold["%s_call%d_block" % (case, i)] = np.stack(
                        [np.broadcast_to(a, args[0].shape) for a in args]
                        + [np.broadcast_to(kw_["companion_fluxratio"], args[0].shape)])
                    gold["%s_call%d_is_host" % (case, i)] = np.array([bool(kw_.get("companion_is_host", False))])
                    gold["%s_call%d_out" % (case, i)] = outv
            else:
                gold[case + "_ncalls"] = np.array([len(captured)])
    gold["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "lnz_cases.npz"), **gold)

    # ---- (3b) the four lnZ_* that calc_probs never calls (API surface) --------
    extra = {"time": t, "flux": f, "sigma": np.array([sigma])}
    ex_cases = []
    ex_plan = [
        ("NTPu_par", lambda: rml.lnZ_NTP_unknown(*b, 3.3, 14.0, tri_path, N, True)),
        ("NTPu_serial", lambda: rml.lnZ_NTP_unknown(*b, 3.3, 14.0, tri_path, 300, False)),
        ("NTPu_empty", lambda: rml.lnZ_NTP_unknown(*b, 3.3, 30.0, tri_path, N, True)),
        ("NEBu_par", lambda: rml.lnZ_NEB_unknown(*b, 3.3, 14.0, tri_path, N, True)),
        ("NEBu_serial", lambda: rml.lnZ_NEB_unknown(*b, 3.3, 14.0, tri_path, 300, False)),
        ("NEBu_empty", lambda: rml.lnZ_NEB_unknown(*b, 3.3, 30.0, tri_path, N, True)),
        ("NTPe_par", lambda: rml.lnZ_NTP_evolved(*b, 3.3, 3.2, 4900.0, 0.0, N, True)),
        ("NTPe_serial", lambda: rml.lnZ_NTP_evolved(*b, 3.3, 3.2, 4900.0, 0.0, 300, False)),
        ("NEBe_par", lambda: rml.lnZ_NEB_evolved(*b, [3.0, 3.6], 3.2, 4900.0, 0.0, N, True)),
        ("NEBe_serial", lambda: rml.lnZ_NEB_evolved(*b, 3.3, 3.2, 4900.0, 0.0, 300, False)),
    ]
    seed = 5000
    for case, fn in ex_plan:
        seed += 1
        np.random.seed(seed)
        del captured[:], logws[:]
        res = fn()
        ex_cases.append(case)
        dicts = res if isinstance(res, tuple) else (res,)
        extra[case + "_seed"] = np.array([seed])
        extra[case + "_nres"] = np.array([len(dicts)])
        for i, d in enumerate(dicts):
            extra["%s_lnZ%d" % (case, i)] = np.array([d["lnZ"]])
            extra["%s_keys%d" % (case, i)] = np.array(sorted(d.keys()))
            for k, v in d.items():
                if k != "lnZ":
                    extra["%s_res%d_%s" % (case, i, k)] = np.atleast_1d(np.asarray(v, dtype=float))
        for i, lw in enumerate(logws):
            extra["%s_logw%d" % (case, i)] = lw
        if case.endswith("_par"):
            for i, (cname, args, kw_, outv) in enumerate(captured):
                shape = max((np.shape(a) for a in args), key=len)
                extra["%s_call%d_name" % (case, i)] = np.array([cname])
                extra["%s_call%d_block" % (case, i)] = np.stack(
                    [np.broadcast_to(a, shape) for a in args]
                    + [np.broadcast_to(kw_["companion_fluxratio"], shape)])
                extra["%s_call%d_out" % (case, i)] = outv
    extra["cases"] = np.array(ex_cases)
    np.savez_compressed(os.path.join(HERE, "lnz_extra.npz"), **extra)

    # ---- (4) BASELINE config 1: TOI-1228, TP scenario, N = 1e4 -------------------------------
    tb, fb, sg = binned_toi1228()
    np.random.seed(20260424 % (2 ** 32))
    del captured[:], logws[:]
    # TIC 300038935: mass, radius, Teff from the stars table shown in TSCIII_tutorial.ipynb cell 14
    res = rml.lnZ_TTP(tb, fb, sg, 29.04992, 2.13, 1.79626, 8557.0, 0.0, 10000, True)
    np.savez_compressed(os.path.join(HERE, "toi1228_ttp.npz"), time=tb, flux=fb,
                        sigma=np.array([sg]), P_orb=np.array([29.04992]),
                        star=np.array([2.13, 1.79626, 8557.0, 0.0]), N=np.array([10000]),
                        seed=np.array([20260424]), lnZ=np.array([res["lnZ"]]),
                        logw=logws[0], block=np.stack(captured[0][1] + [np.zeros_like(captured[0][1][0])]),
                        out=captured[0][3],
                        **{"res_" + k: np.asarray(v, dtype=float) for k, v in res.items() if k != "lnZ"})
    # ---- (5) the reference's own target.calc_depths / calc_probs -------------------------------
    import pandas as pd
    rtr = import_reference_target()
    # the lnZ_* were star-imported into rtr at import time, before the capture wrappers existed
    write_molusc(os.path.join(HERE, "molusc_synth.csv"), np.random.default_rng(5))

    def stars_table():
        return pd.DataFrame({
            "ID": [111, 222, 333, 444], "Tmag": [10.4, 13.0, 15.5, 12.2],
            "Jmag": [9.5, 12.1, 14.6, 11.5], "Hmag": [9.1, 11.7, 14.2, 11.1],
            "Kmag": [9.0, 11.6, 14.1, 11.0], "ra": [10.0, 10.004, 10.01, 9.99],
            "dec": [-5.0, -5.003, -5.01, -4.995], "mass": [0.82, 0.6, np.nan, 1.1],
            "rad": [0.8, 0.58, np.nan, 1.3], "Teff": [5100.0, 4000.0, np.nan, 6000.0],
            "plx": [14.2, 3.0, np.nan, 2.0]})

    def ref_target(stars):
        tg = object.__new__(rtr.target)
        tg.ID, tg.mission, tg.sectors, tg.search_radius, tg.N_pix = 111, "TESS", np.array([1]), 10, 22
        tg.stars, tg.trilegal_fname, tg.trilegal_url = stars, tri_path, None
        tg.pix_coords = [np.array([[10.2, 10.7], [11.9, 11.3], [14.0, 7.5], [8.4, 12.6]]),
                         np.array([[10.6, 10.1], [12.2, 10.9], [14.5, 7.0], [8.9, 12.0]])]
        return tg

    cp = {}
    tg = ref_target(stars_table())
    aps = [np.array([[x, y] for x in range(9, 12) for y in range(10, 13)]),
           np.array([[x, y] for x in range(9, 13) for y in range(9, 12)])]
    tg.calc_depths(0.007, aps)
    cp["depths_fluxratio"], cp["depths_tdepth"] = tg.stars["fluxratio"].values, tg.stars["tdepth"].values
    for i, a_ in enumerate(aps):
        cp["aperture%d" % i] = a_
    tg5 = ref_target(stars_table())
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        tg5.calc_depths(0.007)                        # default 5x5 apertures
    cp["depths5_fluxratio"], cp["depths5_tdepth"] = tg5.stars["fluxratio"].values, tg5.stars["tdepth"].values
    runs = {
        "cc": dict(contrast_curve_file=cc_path, filt="J", drop_scenario=[]),
        "molusc": dict(contrast_curve_file=None, filt="TESS", drop_scenario=["DEB", "BTP"],
                       molusc_file=os.path.join(HERE, "molusc_synth.csv")),
    }
    for rname, kw in runs.items():
        np.random.seed(777)
        with contextlib.redirect_stdout(io.StringIO()):
            tg.calc_probs(t, f, sigma, 3.3, N=1500, parallel=True, verbose=0, **kw)
        pr = tg.probs
        cp[rname + "_lnZ"], cp[rname + "_prob"] = np.array(tg.lnZ), pr["prob"].values
        cp[rname + "_FPP"], cp[rname + "_NFPP"] = np.array([tg.FPP]), np.array([tg.NFPP])
        cp[rname + "_scenario"] = np.array(list(pr["scenario"]))
        cp[rname + "_ID"], cp[rname + "_star_num"] = pr["ID"].values, np.array(tg.star_num)
        for col in ("M_s", "R_s", "P_orb", "inc", "b", "ecc", "w", "R_p", "M_EB", "R_EB"):
            cp[rname + "_" + col] = pr[col].values
        cp[rname + "_u1"], cp[rname + "_u2"] = np.array(tg.u1), np.array(tg.u2)
        cp[rname + "_fluxratio_EB"], cp[rname + "_fluxratio_comp"] = (np.array(tg.fluxratio_EB),
                                                                      np.array(tg.fluxratio_comp))
    cp["time"], cp["flux"], cp["sigma"] = t, f, np.array([sigma])
    np.savez_compressed(os.path.join(HERE, "calc_probs.npz"), **cp)

    # ---- (6) TOI-1228 (BASELINE north-star case): the reference's calc_probs on the tutorial's
    # light curve, contrast curve and star table (examples/TSCIII_tutorial.ipynb cells 5, 7, 18, 20;
    # the TRILEGAL table is synthetic and there is no MOLUSC file: neither is in the tree)
    tb, fb, sg = binned_toi1228()
    t1228 = object.__new__(rtr.target)
    t1228.ID, t1228.mission, t1228.sectors = 300038935, "TESS", np.array([1])
    t1228.search_radius, t1228.N_pix, t1228.trilegal_fname, t1228.trilegal_url = 10, 22, tri_path, None
    t1228.stars = toi1228_stars()
    np.random.seed(1228)
    with contextlib.redirect_stdout(io.StringIO()):
        t1228.calc_probs(tb, fb, sg, 29.04992, contrast_curve_file=os.path.join(HERE, "toi1228_cc.csv"),
                         filt="TESS", N=4000, parallel=True, verbose=0)
    np.savez_compressed(os.path.join(HERE, "toi1228_calc_probs.npz"), time=tb, flux=fb,
                        sigma=np.array([sg]), lnZ=np.array(t1228.lnZ), prob=t1228.probs["prob"].values,
                        FPP=np.array([t1228.FPP]), NFPP=np.array([t1228.NFPP]),
                        scenario=np.array(list(t1228.probs["scenario"])), N=np.array([4000]),
                        seed=np.array([1228]))

    # ---- (7) caller-side helpers of target: star-table edits and the best-fit curves of plot_fits
    # (the reference keeps star IDs as strings; plot_fits matches on str(ID))
    ops = {}
    st = stars_table()
    st["ID"] = st["ID"].astype(str)
    tg = ref_target(st)
    tg.calc_depths(0.007, aps)
    np.random.seed(99)
    with contextlib.redirect_stdout(io.StringIO()):
        tg.calc_probs(t, f, sigma, 3.3, N=1500, parallel=True, verbose=0, contrast_curve_file=cc_path,
                      drop_scenario=["PEB"])
    import matplotlib.pyplot as plt
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        tg.plot_fits(t, f, sigma, save=True, fname=os.path.join(tmp, "fits"))
    panels = plt.gcf().axes
    ops["fit_model"] = np.array([[ln for ln in p.lines if ln.get_linewidth() == 3][0].get_ydata()
                                 for p in panels])
    ops["fit_model_time"] = np.asarray([ln for ln in panels[0].lines if ln.get_linewidth() == 3][0].get_xdata())
    ops["fit_data"] = np.array([[ln for ln in p.lines if ln.get_linewidth() != 3][0].get_ydata()
                                for p in panels])
    ops["fit_labels"] = np.array([[a.get_text() for a in p.texts] for p in panels])
    plt.close("all")
    ops["fit_seed"] = np.array([99])
    tg.add_star(555, 14.5, True)
    tg.add_star(666, 16.0, False)
    ops["add_ID"], ops["add_Tmag"], ops["add_plx"] = (np.array(list(tg.stars["ID"])),
                                                      tg.stars["Tmag"].values.copy(),
                                                      tg.stars["plx"].values.copy())
    ops["add_mass"] = tg.stars["mass"].values.copy()      # update_star below writes in place
    ops["add_pix0"], ops["add_pix1"] = tg.pix_coords
    tg.update_star(666, "mass", 0.4)
    tg.update_star(222, "Teff", 4100.0)
    ops["upd_mass"], ops["upd_Teff"] = tg.stars["mass"].values.copy(), tg.stars["Teff"].values.copy()
    tg.remove_star(np.array([333, 555]))
    ops["rm_ID"], ops["rm_index"] = np.array(list(tg.stars["ID"])), tg.stars.index.values
    tg.remove_star(444)
    ops["rm2_ID"] = np.array(list(tg.stars["ID"]))
    # the two caller-less helpers of funcs.py
    vk = np.array([[12.0, 10.5], [9.3, 8.8], [15.0, 9.5], [14.0, 8.0], [11.0, 5.9]])
    ops["cteff_in"], ops["cteff_out"] = vk, np.array([rfuncs.color_Teff_relations(v, k) for v, k in vk])
    gx, gy = np.linspace(3, 7, 9), np.linspace(2, 6.5, 7)
    ops["gauss_x"], ops["gauss_y"] = gx, gy
    ops["gauss_grid"] = rfuncs.Gauss2D(gx, gy, 5.3, 4.7, 0.75, 2.5)
    ops["gauss_scalar"] = np.array([rfuncs.Gauss2D(4.9, 5.2, 5.3, 4.7, 0.75, 2.5)])
    np.savez_compressed(os.path.join(HERE, "target_ops.npz"), **ops)
    toi465(rtr, tri_path)
    print("wrote", sorted(os.listdir(HERE)))


# ---- (8) BASELINE config 3: TOI-465.01 (examples/example.ipynb cells 3-18) --------------------
def toi465_stars():
    """the 26 stars of examples/example.ipynb cell 7 (notebook output, after calc_depths(0.005))"""
    import pandas as pd
    nan = np.nan
    return pd.DataFrame({
        "ID": [270380593, 270380591, 514519134, 270380594, 630359580, 630359579, 270380595, 630359572,
               270380592, 630359577, 630359570, 270380590, 630359578, 630359576, 630359569, 630359581,
               630359575, 630359589, 270380600, 630359574, 630359571, 630359568, 630359683, 270380588,
               630359682, 270380599],
        "Tmag": [10.7307, 20.0711, 19.7713, 16.0568, 19.8256, 18.7953, 16.7050, 20.3657, 16.7702,
                 19.3124, 19.4021, 16.7147, 20.3807, 20.5930, 20.7690, 19.0413, 20.2057, 19.4333,
                 18.2569, 20.7012, 20.5628, 19.7183, 20.6261, 16.9112, 19.0436, 19.6075],
        "Jmag": [9.906, 16.829, nan, 14.576, nan, nan, 15.478, nan, 15.909, nan, nan, 15.491, nan, nan,
                 nan, nan, nan, nan, 16.575, nan, nan, nan, nan, 16.083, nan, 16.950],
        "Hmag": [9.473, 16.420, nan, 13.973, nan, nan, 15.022, nan, 15.495, nan, nan, 14.853, nan, nan,
                 nan, nan, nan, nan, 16.091, nan, nan, nan, nan, 15.432, nan, 16.386],
        "Kmag": [9.339, 15.772, nan, 13.765, nan, nan, 14.588, nan, 15.531, nan, nan, 14.608, nan, nan,
                 nan, nan, nan, nan, 15.503, nan, nan, nan, nan, 15.354, nan, 15.660],
        "ra": [32.781765, 32.780541, 32.780333, 32.770020, 32.785538, 32.768955, 32.804587, 32.757731,
               32.809521, 32.760939, 32.780396, 32.787580, 32.760596, 32.814931, 32.807530, 32.784037,
               32.817551, 32.813509, 32.763755, 32.737279, 32.750030, 32.821061, 32.738867, 32.819544,
               32.741726, 32.737178],
        "dec": [2.418021, 2.404015, 2.403886, 2.426293, 2.431950, 2.433112, 2.428540, 2.402821,
                2.411144, 2.437847, 2.389176, 2.388617, 2.440020, 2.421735, 2.396169, 2.454356,
                2.426406, 2.439309, 2.454164, 2.418009, 2.383120, 2.390708, 2.446400, 2.380961,
                2.453939, 2.452297],
        "mass": [0.811, nan, nan, 0.513011, nan, nan, 0.64, nan, 0.95, nan, nan, 0.668646, nan, nan, nan,
                 0.45, nan, nan, nan, nan, nan, nan, nan, 0.85, nan, nan],
        "rad": [0.847380, nan, nan, 0.515342, nan, nan, 0.795370, nan, 0.534628, nan, nan, 0.707274, nan,
                nan, nan, 0.385007, nan, nan, nan, nan, nan, nan, nan, 0.496960, nan, nan],
        "Teff": [4936.0, nan, nan, 3516.0, nan, nan, 4073.0, nan, 5439.0, 3750.0, nan, 4016.0, nan, nan,
                 nan, 3479.0, nan, nan, 3874.0, nan, nan, nan, nan, 5080.0, 3583.0, nan],
        "plx": [8.163660, nan, nan, 2.251200, -0.477026, 0.087234, 0.839486, -1.642620, 0.595253,
                1.619320, 0.345174, 0.894513, nan, nan, 1.620700, -0.127052, 1.092730, 2.080520,
                0.801983, nan, nan, 0.089077, nan, 0.683115, 0.907077, nan],
        "fluxratio": [9.986416e-01, 2.904046e-05, 3.615892e-05, 1.252310e-03, 2.507577e-05,
                      1.156098e-05, 3.595686e-06, 2.355716e-09, 6.261372e-07, 6.005195e-09,
                      2.524618e-09, 7.237678e-09, 3.974583e-10, 3.570753e-11, 3.284628e-11,
                      1.842151e-13, 3.791387e-13, 1.958695e-13, 7.389879e-15, 1.503412e-19,
                      1.765258e-21, 4.631424e-20, 4.447183e-26, 3.037689e-24, 2.707183e-27,
                      1.995252e-30],
        "tdepth": [0.005007] + [0.0] * 25})


def toi465_blend_stars():
    """BASELINE config 3's shape, "20 contaminating stars": the target and its 20 nearest neighbours
    of the table above with SYNTHETIC aperture flux ratios (target 0.8, every neighbour 0.01), so that
    all 20 neighbours spawn NTP/NEB/NEBx2P scenarios: 75 scenarios.  (In the real field no neighbour
    is bright enough to host a 5000 ppm signal and the notebook's run has 15 scenarios.)"""
    st = toi465_stars().iloc[:21].copy()
    st["fluxratio"] = [0.8] + [0.01] * 20
    st["tdepth"] = 0.005 / st["fluxratio"].values
    return st


def binned_toi465(n_bins=100):
    """examples/example.ipynb cell 9: TessLightCurve(...).bin(time_bin_size=(tmax - tmin)/100) and
    flux_err_0 = mean of the binned flux errors (lightkurve: mean flux per bin, error of the mean
    sqrt(sum err^2)/n per bin; bins of fixed width starting at the first time stamp)."""
    import pandas as pd
    lc = pd.read_csv(os.path.join(REF, "examples", "TOI465_01_lightcurve.csv"), header=None)
    t, y, e = lc[0].values.astype(float), lc[1].values.astype(float), lc[2].values.astype(float)
    width = (t.max() - t.min()) / n_bins
    idx = np.minimum(((t - t.min()) / width).astype(int), n_bins - 1)
    tb = np.array([t.min() + (i + 0.5) * width for i in range(n_bins) if np.any(idx == i)])
    fb = np.array([y[idx == i].mean() for i in range(n_bins) if np.any(idx == i)])
    eb = np.array([np.sqrt(np.sum(e[idx == i] ** 2)) / np.sum(idx == i) for i in range(n_bins)
                   if np.any(idx == i)])
    return (t, y, e), tb, fb, float(np.mean(eb))


def toi465(rtr, tri_path):
    """the reference's own calc_probs on TOI-465.01 with its contrast curve (example.ipynb cell 18),
    (a) the notebook's star table (15 scenarios) and (b) the 1 + 20-star blend (75 scenarios)"""
    import contextlib
    import io
    import shutil
    raw, tb, fb, sg = binned_toi465()
    cc_dst = os.path.join(HERE, "toi465_cc.csv")
    shutil.copyfile(os.path.join(REF, "examples", "TOI465_01_contrastcurve.csv"), cc_dst)   # data file
    out = {"raw_time": raw[0], "raw_flux": raw[1], "raw_flux_err": raw[2], "time": tb, "flux": fb,
           "sigma": np.array([sg]), "P_orb": np.array([3.836169])}
    for tag, stars, N, seed in (("real", toi465_stars(), 3000, 465), ("blend", toi465_blend_stars(), 1500, 4651)):
        tg = object.__new__(rtr.target)
        tg.ID, tg.mission, tg.sectors = 270380593, "TESS", np.array([4])
        tg.search_radius, tg.N_pix, tg.trilegal_fname, tg.trilegal_url = 10, 22, tri_path, None
        tg.stars = stars
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            tg.calc_probs(tb, fb, sg, 3.836169, contrast_curve_file=cc_dst, N=N, parallel=True, verbose=0)
        out[tag + "_lnZ"], out[tag + "_prob"] = np.array(tg.lnZ), tg.probs["prob"].values
        out[tag + "_FPP"], out[tag + "_NFPP"] = np.array([tg.FPP]), np.array([tg.NFPP])
        out[tag + "_scenario"] = np.array(list(tg.probs["scenario"]))
        out[tag + "_ID"] = tg.probs["ID"].values
        out[tag + "_N"], out[tag + "_seed"] = np.array([N]), np.array([seed])
        for col in ("M_s", "R_s", "P_orb", "inc", "b", "ecc", "w", "R_p", "M_EB", "R_EB"):
            out[tag + "_" + col] = tg.probs[col].values
        for c in stars.columns:
            out[tag + "_stars_" + c] = stars[c].values.astype(float)
    np.savez_compressed(os.path.join(HERE, "toi465_calc_probs.npz"), **out)


def write_molusc(path, rng, n=900):
    """synthetic MOLUSC output with the three columns marginal_likelihoods.py:458-462 reads"""
    import pandas as pd
    pd.DataFrame({"semi-major axis(AU)": 10 ** rng.uniform(0, 3, n),
                  "eccentricity": rng.uniform(0, 0.9, n),
                  "mass ratio": rng.uniform(0.02, 1.0, n)}).to_csv(path, index=False)


def _with_N(plan, name, rml, b, s, tri_path, n, P, kw):
    """serial-path call with a smaller N (the reference loops over draws in Python)"""
    cc, filt = kw.get("cc"), kw.get("filt", "TESS")
    base = (*b, P, s["M_s"], s["R_s"], s["Teff"])
    mags = (s["Tmag"], s["Jmag"], s["Hmag"], s["Kmag"], tri_path, cc, filt, n, False)
    bound = (s["Z"], s["plx"], cc, filt, n, False)
    table = {
        "TTP": lambda: rml.lnZ_TTP(*base, s["Z"], n, False),
        "TEB": lambda: rml.lnZ_TEB(*base, s["Z"], n, False),
        "PTP": lambda: rml.lnZ_PTP(*base, *bound), "PEB": lambda: rml.lnZ_PEB(*base, *bound),
        "STP": lambda: rml.lnZ_STP(*base, *bound), "SEB": lambda: rml.lnZ_SEB(*base, *bound),
        "DTP": lambda: rml.lnZ_DTP(*base, s["Z"], *mags), "DEB": lambda: rml.lnZ_DEB(*base, s["Z"], *mags),
        "BTP": lambda: rml.lnZ_BTP(*base, *mags), "BEB": lambda: rml.lnZ_BEB(*base, *mags),
    }
    return table[name]()




def toi1228_stars():
    """the six stars with tdepth > 0 of examples/TSCIII_tutorial.ipynb cell 18 (notebook output)"""
    import pandas as pd
    return pd.DataFrame({
        "ID": [300038935, 300038933, 300038940, 300038932, 300038925, 300038947],
        "Tmag": [9.0963, 14.2544, 14.8737, 17.0169, 14.2296, 12.4406],
        "Jmag": [8.887, 13.082, 13.832, 16.356, 13.282, 11.452],
        "Hmag": [8.854, 12.418, 13.213, 15.803, 12.879, 10.912],
        "Kmag": [8.823, 12.225, 13.137, 15.684, 12.705, 10.810],
        "ra": [107.843696, 107.852043, 107.848770, 107.860272, 107.852177, 107.874142],
        "dec": [-68.833491, -68.832404, -68.839563, -68.829404, -68.817218, -68.852895],
        "mass": [2.13, 0.58456, 0.75, 0.96, 0.88, np.nan],
        "rad": [1.79626, 0.595692, 0.641739, 0.580447, 0.863853, 3.22447],
        "Teff": [8557.0, 3922.0, 4690.0, 5484.0, 5192.0, 4986.0],
        "plx": [3.64491, 3.70654, 1.93455, 0.565248, 1.5691, 1.04073],
        "fluxratio": [0.979954, 0.008361, 0.004675, 0.000589, 0.001471, 0.003010],
        "tdepth": [0.000415, 0.048680, 0.087064, 0.690429, 0.276603, 0.135210]})


if __name__ == "__main__":
    main()
