"""Which draws make set_sampling("numpy-device") differ from set_sampling("numpy") (= the reference's host arithmetic)?
One lnZ_* function, the notebook inputs of TOI-465.01 with the contrast curve, N = 1e6, a few seeds: per-draw masks,
priors and columns of the host path (captured at marginal_likelihoods._evidence) against the draw kernel's (fused.DUMP).
    python profiles/r05/diag_numpy_device_vs_numpy.py DTP 5"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import triceratops_amd  # noqa: E402
from triceratops_amd import fused, marginal_likelihoods as ml  # noqa: E402
import anchors  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "DTP"
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000
stars, t, f, sigma, P = anchors.inputs("toi465_cc")
s0 = stars.iloc[0]
args = (t, f, sigma, P, float(s0["mass"]), float(s0["rad"]), float(s0["Teff"]), 0.0, float(s0["Tmag"]), float(s0["Jmag"]),
        float(s0["Hmag"]), float(s0["Kmag"]), anchors.TRILEGAL, anchors.CC465, "TESS", N, True)
if name in ("PTP", "STP", "PEB", "SEB"):
    args = (t, f, sigma, P, float(s0["mass"]), float(s0["rad"]), float(s0["Teff"]), 0.0, float(s0["plx"]), anchors.CC465, "TESS", N, True)
for seed in range(1001, 1001 + nseeds):
    captured = {}
    orig = ml._evidence

    def spy(model, is_host, parallel, time, flux, sigma_, cols, mask, lnprior, N_, exptime, nsamples):
        captured.setdefault("calls", []).append((cols, mask.copy(), None if lnprior is None else lnprior.copy()))
        return orig(model, is_host, parallel, time, flux, sigma_, cols, mask, lnprior, N_, exptime, nsamples)

    ml._evidence = spy
    triceratops_amd.set_sampling("numpy")
    np.random.seed(seed)
    host = getattr(ml, "lnZ_" + name)(*args)
    ml._evidence = orig
    triceratops_amd.set_sampling("numpy-device")
    fused.DUMP = []
    np.random.seed(seed)
    dev = getattr(ml, "lnZ_" + name)(*args)
    dump = fused.DUMP[0]
    fused.DUMP = None
    hd = host if isinstance(host, tuple) else (host,)
    dd = dev if isinstance(dev, tuple) else (dev,)
    print("seed %d: lnZ host %s device %s diff %s" % (seed, [d["lnZ"] for d in hd], [d["lnZ"] for d in dd],
                                                       ["%.2e" % abs(a["lnZ"] - b["lnZ"]) for a, b in zip(hd, dd)]))
    cols, mask, lnprior = captured["calls"][0]
    dmask = dump["mask"].cpu().numpy().astype(bool)
    diff = np.flatnonzero(mask != dmask)
    print("   masked draws host %d device %d; masks differ at %d draws %s" % (mask.sum(), dmask.sum(), diff.size, diff[:8]))
    if lnprior is not None:
        dl = dump["lnprior"].cpu().numpy()
        both = mask & dmask
        d = np.abs(lnprior[both] - dl[both])
        d[np.isinf(lnprior[both]) & np.isinf(dl[both])] = 0.0
        worst = np.argsort(np.nan_to_num(d, nan=np.inf, posinf=np.inf))[::-1][:5]
        print("   lnprior of the masked draws: max |host - device| %.3e; infinities host %d device %d" %
              (np.nanmax(d), np.isinf(lnprior[both]).sum(), np.isinf(dl[both]).sum()))
        ii = np.flatnonzero(both)[worst]
        for i in ii:
            print("      draw %d: host %.17g device %.17g" % (i, lnprior[i], dl[i]))
    dcols = dump["cols"].cpu().numpy()
    for j, c in enumerate(cols):
        if isinstance(c, np.ndarray):
            both = mask & dmask
            r = np.abs(c[both] - dcols[j][both]) / np.maximum(np.abs(c[both]), 1e-300)
            print("   column %d: max relative difference over the masked draws %.2e" % (j, np.nanmax(r)))
