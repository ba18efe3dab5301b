#!/usr/bin/env python3
"""Pin the transit arithmetic to pytransit itself, wherever pytransit==2.2 imports.

The build container and the GPU box have neither pytransit nor numba (no wheel, no network), so the fixtures of
make_golden.py were made with the CPU oracle standing at the reference's pytransit seam (likelihoods.py:15, 24-25):
Kepler + Mandel-Agol + supersampling are pinned to quadrature and to the notebooks only.  This script is the other
half, ready to run: on any machine where `import pytransit` works it evaluates
`pytransit.QuadraticModel(interpolate=False)` -- the constructor and the three calls the reference makes
(likelihoods.py:24-25, 61-71, 348-349, 414-422) -- on

  (i)   the edge rows of tests/test_gpu_kernels.py::_raw_stress_rows (same generator, same seed),
  (ii)  every parameter block the imported reference handed to lnL_*_p in tests/golden/lnz_cases.npz, converted
        to pytransit's (k, t0, p, a, i, e, w) + (u1, u2) exactly as likelihoods.py:337-349, 399-422 does
        (primary eclipse on the light curve's stamps; the EB secondaries on linspace(-0.05, 0.05, 25), no supersampling),
  (iii) TOI-1228's lnZ_TTP block (tests/golden/toi1228_ttp.npz),

for nsamples in {1, 20}, and writes tests/golden/pytransit_pin.npz (inputs, fluxes, pytransit's version).
tests/test_pytransit_pin.py then compares the oracle (CPU) and the HIP kernels (-m gpu) with it; it is skipped only
while the file is absent.  Only DATA leaves this script.

    python tests/golden/make_pytransit_pin.py            # needs pytransit; nothing else of the reference
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pytransit_pin.npz")

# astropy's cgs constants as the reference reads them (likelihoods.py:17-22)
RSUN, REARTH = 69570000000.0, 637810000.0
EXPTIME = 0.00139
NSAMPLES = (1, 20)
MAX_ROWS = 400            # rows kept per block (the file stays small; the rows are the block's first)


def raw_stress_rows(rng, n):
    """the generator of tests/test_gpu_kernels.py::_raw_stress_rows (kept in step by tests/test_pytransit_pin.py)"""
    k = np.where(rng.random(n) < 0.7, rng.uniform(0.01, 0.3, n), rng.uniform(0.3, 1.5, n))
    a = 10 ** rng.uniform(np.log10(1.5), np.log10(60), n)
    e = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 0.95, n))
    w = rng.uniform(0, 2 * np.pi, n)
    b = rng.uniform(0, 1 + k)
    inc = np.arccos(np.clip(b / (a * (1 - e * e) / (1 + e * np.sin(w))), 0, 1))
    per = 10 ** rng.uniform(np.log10(0.3), 2, n)
    rows = np.stack([k, rng.uniform(-0.02, 0.02, n), per, a, inc, e, w, rng.uniform(0.1, 0.6, n),
                     rng.uniform(0.05, 0.4, n)])
    return np.ascontiguousarray(rows[:, a * (1 - e) > 1 + k])


def k_rule(k):
    """likelihoods.py:405-406 / 417-418 (the vector rule, without abs)"""
    k = np.array(k, dtype=float, copy=True)
    k[(k - 1.0) < 1e-6] *= 0.999
    return k


def block_to_pv(name, block):
    """A lnL_*_p argument block (the reference's argument order + companion_fluxratio last) -> list of
    (tag, pvp (n,7), ldc (n,2), secondary?) as the reference builds them (likelihoods.py:337-349, 399-422)."""
    z = np.zeros(block.shape[1])
    if name == "lnL_TP_p":
        R_p, P, inc, a, R_s, u1, u2, ecc, argp = block[:9]
        k = R_p * REARTH / (R_s * RSUN)
        pv = np.stack([k, z, P, a / (R_s * RSUN), inc * np.pi / 180.0, ecc, (90.0 - argp) * np.pi / 180.0], axis=1)
        return [("primary", pv, np.stack([u1, u2], axis=1), False)]
    R_EB, _fr, P, inc, a, R_s, u1, u2, ecc, argp = block[:10]
    a_R, i_r = a / (R_s * RSUN), inc * np.pi / 180.0
    ldc = np.stack([u1, u2], axis=1)
    pv1 = np.stack([k_rule(R_EB / R_s), z, P, a_R, i_r, ecc, (90.0 - argp) * np.pi / 180.0], axis=1)
    out = [("primary", pv1, ldc, False)]
    if name == "lnL_EB_p":
        pv2 = np.stack([k_rule(R_s / R_EB), z, P, a_R, i_r, ecc, (90.0 - argp + 180.0) * np.pi / 180.0], axis=1)
        out.append(("secondary", pv2, ldc, True))
    return out


def collect_inputs():
    """[(tag, time, exptime, pvp, ldc, nsamples tuple)]"""
    items = []
    rng = np.random.default_rng(20260424 + 21)
    rows = raw_stress_rows(rng, 1200)[:, :MAX_ROWS]
    t = np.linspace(-0.25, 0.25, 200)
    for tag, expt in (("stress_short", EXPTIME), ("stress_long", 0.0204)):
        items.append((tag, t, expt, rows[:7].T.copy(), rows[7:9].T.copy(), NSAMPLES))
    g = np.load(os.path.join(HERE, "lnz_cases.npz"), allow_pickle=False)
    time = g["time"]
    for case in g["cases"]:
        i = 0
        while "%s_call%d_block" % (case, i) in g.files:
            name = str(g["%s_call%d_name" % (case, i)][0])
            block = g["%s_call%d_block" % (case, i)][:, :MAX_ROWS]
            if block.shape[1] and name.endswith("_p"):
                for part, pv, ldc, sec in block_to_pv(name, block):
                    if sec:
                        items.append(("%s_call%d_%s" % (case, i, part), np.linspace(-0.05, 0.05, 25), 0.0, pv, ldc, (1,)))
                    else:
                        items.append(("%s_call%d_%s" % (case, i, part), time, EXPTIME, pv, ldc, NSAMPLES))
            i += 1
    t1228 = np.load(os.path.join(HERE, "toi1228_ttp.npz"), allow_pickle=False)
    for part, pv, ldc, _ in block_to_pv("lnL_TP_p", t1228["block"][:, :MAX_ROWS]):
        items.append(("toi1228_ttp_" + part, t1228["time"], EXPTIME, pv, ldc, NSAMPLES))
    return items


def main():
    try:
        import pytransit
        from pytransit import QuadraticModel
    except Exception as exc:                    # noqa: BLE001
        print("pytransit does not import here (%s: %s): nothing written.  Run this script where "
              "`pip install pytransit==2.2` is possible and commit tests/golden/pytransit_pin.npz."
              % (type(exc).__name__, exc))
        return 2
    out = {"pytransit_version": np.array([getattr(pytransit, "__version__", "unknown")]),
           "exptime_default": np.array([EXPTIME])}
    tags = []
    for tag, time, expt, pv, ldc, nss in collect_inputs():
        tags.append(tag)
        out[tag + "_time"] = time
        out[tag + "_exptime"] = np.array([expt])
        out[tag + "_pvp"] = pv
        out[tag + "_ldc"] = ldc
        for ns in nss:
            tm = QuadraticModel(interpolate=False)                       # likelihoods.py:24-25
            if ns == 1 and expt == 0.0:
                tm.set_data(time)                                        # :135, :421 (tm_sec)
            else:
                tm.set_data(time, exptimes=expt, nsamples=ns)            # :61, :348, :414
            out["%s_flux_ns%d" % (tag, ns)] = np.asarray(tm.evaluate_pv(pv, ldc), dtype=np.float64)    # :349, :415, :422
        # the scalar call shape on the block's first row (:62-71)
        tm = QuadraticModel(interpolate=False)
        tm.set_data(time, exptimes=expt, nsamples=nss[-1]) if expt > 0.0 else tm.set_data(time)
        r = pv[0]
        out[tag + "_flux_ps"] = np.asarray(tm.evaluate_ps(r[0], ldc[0], r[1], r[2], r[3], r[4], r[5], r[6]), dtype=np.float64)
    out["tags"] = np.array(tags)
    np.savez_compressed(OUT, **out)
    print("wrote %s: %d blocks, pytransit %s" % (OUT, len(tags), out["pytransit_version"][0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
