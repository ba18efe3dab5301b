"""Where could the notebooks and this implementation part ways?  One input changed at a time.

examples/example.ipynb cell 25 (TOI-411.02) printed TP : PTP : STP = 0.751 : 0.119 : 0.0338 from ONE run at N = 1e6
with the real pytransit; this implementation gives 0.797 : 0.156 : 0.0448 (300 runs), i.e. ln(PTP/TP) is 0.21 higher
here at a per-run scatter of 0.07; cell 14 (TOI-465.01, no contrast curve) printed FPP = 0.0432 +- 0.0578 over 20
runs against 0.005 here.  The reference's CURRENT code run on the CPU with the oracle at pytransit's seam agrees with
this implementation (tests/golden/reference_runs.npz), so what is left is (a) an input this repository had to make up
because the notebook's came from a package that is not here (lightkurve's binning, the sigma it reports, the digits of
the printed star table), (b) the release that made the notebooks, (c) pytransit's own arithmetic.  This script
measures (a): for TOI-411.02 and TOI-465.01 (no contrast curve) every such input is changed ALONE and the shifts of

    ln(PTP/TP), ln(STP/TP), FPP

are reported as paired differences against the baseline over the same seeds (device sampling: a draw's random numbers
depend on seed and index only, so a variant and the baseline share their draws and the difference is far less noisy
than either), mean +- standard error.

    python profiles/anchor_sensitivity.py [n_seeds=100] > profiles/r04/anchor_sensitivity.txt

Variants
  bin:*     the bin-edge conventions lightkurve / astropy may have used (lightkurve is not in this image): the last
            stamp (== the last edge) kept in the last bin (baseline), dropped, or in a 101st bin of its own; bins
            centred on 0; the reported time = bin centre (baseline), left edge, mean stamp of the bin; nan-mean =
            mean (no NaN in the files) is not listed
  sigma:*   flux_err_0: mean of the binned errors (baseline, what the cells compute), x0.9 / x0.95 / x1.05 / x1.1,
            median of the binned errors, the scatter of the out-of-transit binned points, the 101-bin mean
  star:*    +-1 in the last printed digit of each typed-in star-table value of the target (mass, rad, Teff, plx, Tmag)
  P:*       +-1 in the last digit of P_orb
  mode:*    numpy-device / numpy sampling (the reference's own generator and draw order) against device
  parallel  parallel=False (TOI-465.01's cell 9 run; the per-draw-loop semantics of App. C)
  expo:*    exptime / nsamples: no supersampling, 10 and 40 sub-exposures, a 30-minute exposure
  N:*       1e5 and 1e7 draws (bias of ln(mean) at finite N)
  prior:*   (round 5) the bound-companion prior: the earlier forms the reference keeps as comments (priors.py:661-688,
            749-776, 863-890, 951-970) and the separation limit used when no contrast curve is given
            (ANCHOR_ONLY=prior python profiles/anchor_sensitivity.py 100 > profiles/r05/anchor_prior_forms.txt)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import anchors  # noqa: E402
from helpers import gold  # noqa: E402

RAW = gold("example_lightcurves.npz")
N_BINS = 100


def bin_raw(raw, last="keep", centred=False, stamp="centre"):
    """fixed-width bins of width (tmax - tmin) / 100 as the notebook cells ask lightkurve for"""
    t, y, e = raw[:, 0], raw[:, 1], raw[:, 2]
    width = (t.max() - t.min()) / N_BINS
    start = -0.5 * width * N_BINS if centred else t.min()
    idx = np.floor((t - start) / width).astype(int)
    n_bins = N_BINS
    if last == "keep":
        idx = np.minimum(idx, N_BINS - 1)
    elif last == "drop":
        keep = idx < N_BINS
        t, y, e, idx = t[keep], y[keep], e[keep], idx[keep]
    elif last == "own":
        n_bins = N_BINS + 1
    idx = np.clip(idx, 0, n_bins - 1)
    used = [i for i in range(n_bins) if np.any(idx == i)]
    if stamp == "centre":
        tb = np.array([start + (i + 0.5) * width for i in used])
    elif stamp == "left":
        tb = np.array([start + i * width for i in used])
    else:
        tb = np.array([t[idx == i].mean() for i in used])
    fb = np.array([y[idx == i].mean() for i in used])
    eb = np.array([np.sqrt(np.sum(e[idx == i] ** 2)) / np.sum(idx == i) for i in used])
    return tb, fb, eb


def base_inputs(case):
    stars, t, f, sigma, P = anchors.inputs(case)
    return dict(stars=stars, time=np.asarray(t, float), flux=np.asarray(f, float), sigma=float(sigma), P=float(P),
                kw={}, sampling="device", N=1_000_000)


def variants(case):
    key = anchors.CASES[case]["key"]
    raw = RAW[key]
    b = base_inputs(case)
    tb, fb, eb = bin_raw(raw)
    # the fixture must be what the baseline binning gives (make_anchors.py / make_golden.py)
    assert np.allclose(tb, b["time"], rtol=0, atol=1e-12) and np.allclose(fb, b["flux"], rtol=0, atol=1e-12)
    assert abs(np.mean(eb) - b["sigma"]) < 1e-12
    out = [("baseline", dict())]

    def lc(name, **kw):
        t, f, e = bin_raw(raw, **kw)
        out.append((name, dict(time=t, flux=f, sigma=float(np.mean(e)))))

    lc("bin:last stamp dropped", last="drop")
    lc("bin:last stamp in a 101st bin", last="own")
    lc("bin:bins centred on 0", centred=True)
    lc("bin:time = left bin edge", stamp="left")
    lc("bin:time = mean stamp of the bin", stamp="mean")
    for fac in (0.9, 0.95, 1.05, 1.1):
        out.append(("sigma:x%.2f" % fac, dict(sigma=b["sigma"] * fac)))
    out.append(("sigma:median of the binned errors", dict(sigma=float(np.median(eb)))))
    oot = np.abs(tb) > 0.6 * np.abs(tb).max()
    out.append(("sigma:std of the out-of-transit bins", dict(sigma=float(np.std(fb[oot], ddof=1)))))
    digits = {"toi411": dict(mass=0.01, rad=1e-5, Teff=1.0, plx=1e-4, Tmag=1e-4),
              "toi465": dict(mass=1e-3, rad=1e-5, Teff=1.0, plx=1e-5, Tmag=1e-4)}[key]
    for col, d in digits.items():
        for sgn in (+1, -1):
            st = b["stars"].copy()
            st.loc[0, col] = st.loc[0, col] + sgn * d
            out.append(("star:%s %+g" % (col, sgn * d), dict(stars=st)))
    for sgn in (+1, -1):
        out.append(("P:%+g" % (sgn * 1e-6), dict(P=b["P"] + sgn * 1e-6)))
    out.append(("mode:numpy-device", dict(sampling="numpy-device")))
    out.append(("mode:numpy", dict(sampling="numpy", seeds=16)))
    out.append(("parallel=False", dict(kw=dict(parallel=False))))
    out.append(("expo:nsamples=1", dict(kw=dict(nsamples=1))))
    out.append(("expo:nsamples=10", dict(kw=dict(nsamples=10))))
    out.append(("expo:nsamples=40", dict(kw=dict(nsamples=40))))
    out.append(("expo:exptime=30 min", dict(kw=dict(exptime=0.0208333))))
    out.append(("N:1e5", dict(N=100_000)))
    out.append(("N:1e7", dict(N=10_000_000, seeds=32)))
    # Round 5: the bound-companion prior itself.  The reference keeps an EARLIER form of it as comments
    # (priors.py:661-688, 749-776: f_comp = t1 + t2 + t3 + t4_partial ... for the planet scenarios; :863-890, 951-970: + t1
    # for the binaries).  Without a contrast curve every draw has the same separation limit (2.2 arcsec at the target's
    # distance: log10 P_max = 5.74 for TOI-411.02, 6.21 for TOI-465.01), so a form of the prior is ONE factor on PTP and STP
    # alike; beyond log10 P = 5.5 the earlier planet form is the current one plus the constant t1 + t2 + t3, which the
    # host-computed t4 of trx_draw_args can carry.  Also: the separation limit that stands in for a contrast curve.
    out.append(("prior:earlier TP form (+ t1 + t2 + t3)", dict(hook="tp_close")))
    out.append(("prior:earlier EB form (+ t1)", dict(hook="eb_t1")))
    out.append(("prior:no-cc separation 1.1 arcsec", dict(no_cc_sep=1.1)))
    out.append(("prior:no-cc separation 4.4 arcsec", dict(no_cc_sep=4.4)))
    return b, out


def _hook(name):
    from triceratops_amd import fused

    def tp_close(a, kind):          # planet scenarios: + t1 + t2 + t3 (valid where log10 P_max >= 5.5: checked in run())
        if kind == fused.PRIOR_BOUND_TP:
            a.t4 = a.t4 + a.f1 + a.t2 + a.t3

    def eb_t1(a, kind):             # binaries: + t1 (every range from log10 P = 2 on carries t2)
        if kind == fused.PRIOR_BOUND_EB:
            a.t2 = a.t2 + a.f1

    return {"tp_close": tp_close, "eb_t1": eb_t1}[name]


def run(case, inp, seed):
    import torch
    import triceratops_amd
    from triceratops_amd.triceratops import target
    c = anchors.CASES[case]
    tg = target(c["ID"], np.array([1]), mission=c["mission"], stars=inp["stars"], trilegal_fname=anchors.TRILEGAL)
    prev = triceratops_amd.get_sampling()
    triceratops_amd.set_sampling(inp["sampling"])
    from triceratops_amd import fused
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        kw = dict(parallel=True)
        kw.update(inp["kw"])
        fused.BOUND_HOOK = _hook(inp["hook"]) if inp.get("hook") else None
        fused.NO_CC_SEPARATION = inp.get("no_cc_sep", 2.2)
        tg.calc_probs(inp["time"], inp["flux"], inp["sigma"], inp["P"], contrast_curve_file=c["cc"], N=inp["N"],
                      verbose=0, **kw)
    finally:
        fused.BOUND_HOOK = None
        fused.NO_CC_SEPARATION = 2.2
        triceratops_amd.set_sampling(prev)
    lnZ = np.array(tg.lnZ)
    i_tp, i_ptp, i_stp = (anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP"))
    return lnZ[i_ptp] - lnZ[i_tp], lnZ[i_stp] - lnZ[i_tp], float(tg.FPP), lnZ[i_tp]


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    cases = sys.argv[2].split(",") if len(sys.argv) > 2 else ["toi411", "toi465_nocc"]
    report = {}
    for case in cases:
        b, var = variants(case)
        only = os.environ.get("ANCHOR_ONLY")           # e.g. ANCHOR_ONLY=prior: the baseline and the prior:* variants
        if only:
            var = [v for v in var if v[0] == "baseline" or v[0].startswith(only)]
        nb_prob, nb_fpp, _ = anchors.notebook(case)
        i_tp, i_ptp, i_stp = (anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP"))
        nb = (np.log(nb_prob[i_ptp] / nb_prob[i_tp]), np.log(nb_prob[i_stp] / nb_prob[i_tp]), nb_fpp)
        res = {}
        t00 = time.perf_counter()
        for name, ov in var:
            inp = dict(b)
            inp.update({k: v for k, v in ov.items() if k != "seeds"})
            n = min(n_seeds, ov.get("seeds", n_seeds))
            t0 = time.perf_counter()
            res[name] = np.array([run(case, inp, 1000 + s) for s in range(n)])
            sys.stderr.write("%s %s: %d runs, %.3f s each\n" % (case, name, n, (time.perf_counter() - t0) / n))
        base = res["baseline"]
        print("==== %s: %d seeds, N = 1e6, device sampling; %d variants, %.0f s" % (case, n_seeds, len(var) - 1,
                                                                                    time.perf_counter() - t00))
        print("baseline (mean +- std of a run): ln(PTP/TP) %.4f +- %.4f   ln(STP/TP) %.4f +- %.4f   FPP %.5f +- %.5f"
              % (base[:, 0].mean(), base[:, 0].std(ddof=1), base[:, 1].mean(), base[:, 1].std(ddof=1),
                 base[:, 2].mean(), base[:, 2].std(ddof=1)))
        print("notebook (one run)             : ln(PTP/TP) %.4f            ln(STP/TP) %.4f            FPP %.5f"
              % nb)
        off = (nb[0] - base[:, 0].mean(), nb[1] - base[:, 1].mean(), nb[2] - base[:, 2].mean())
        print("notebook - baseline            : %+.4f (%.1f sigma of a run)   %+.4f (%.1f sigma)   %+.5f"
              % (off[0], off[0] / base[:, 0].std(ddof=1), off[1], off[1] / base[:, 1].std(ddof=1), off[2]))
        print("%-40s %5s  %-22s %-22s %-24s %-20s" % ("variant (changed ALONE)", "runs", "d ln(PTP/TP)", "d ln(STP/TP)",
                                                       "d FPP", "d lnZ_TP"))
        rows = []
        for name, _ in var[1:]:
            r = res[name]
            n = r.shape[0]
            d = r - base[:n]
            m, se = d.mean(axis=0), d.std(axis=0, ddof=1) / np.sqrt(n)
            print("%-40s %5d  %+8.4f +- %-10.4f %+8.4f +- %-10.4f %+9.5f +- %-11.5f %+8.3f +- %.3f"
                  % (name, n, m[0], se[0], m[1], se[1], m[2], se[2], m[3], se[3]))
            rows.append(dict(variant=name, runs=n, d_ln_ptp_tp=[m[0], se[0]], d_ln_stp_tp=[m[1], se[1]],
                             d_fpp=[m[2], se[2]], d_lnz_tp=[m[3], se[3]]))
        carried = max(rows, key=lambda q: abs(q["d_ln_ptp_tp"][0]) if q["variant"].split(":")[0] not in ("sigma",) or
                      "x" not in q["variant"] else 0.0)
        print("largest single shift of ln(PTP/TP) among the non-scaling variants: %s (%+.4f); the notebook sits %+.4f away"
              % (carried["variant"], carried["d_ln_ptp_tp"][0], off[0]))
        report[case] = dict(baseline=dict(ln_ptp_tp=[base[:, 0].mean(), base[:, 0].std(ddof=1)],
                                          ln_stp_tp=[base[:, 1].mean(), base[:, 1].std(ddof=1)],
                                          fpp=[base[:, 2].mean(), base[:, 2].std(ddof=1)]),
                            notebook=dict(ln_ptp_tp=nb[0], ln_stp_tp=nb[1], fpp=nb[2]), variants=rows)
        sys.stdout.flush()
    if os.environ.get("ANCHOR_JSON"):
        with open(os.environ["ANCHOR_JSON"], "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
