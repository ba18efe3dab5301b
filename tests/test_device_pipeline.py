"""The torch expression of the device-resident scenario pipeline (tests/torch_pipeline.py: round 1's device
path, now the cross-check of the fused draw kernel).

Its samplers, relations and priors are deterministic functions of the uniforms, so they are
checked value for value against the host path (itself bit-identical to the reference).  The whole
pipeline is then run with a random-number source that replays numpy's global stream in the
reference's draw order: with that, every seeded golden case of the imported reference must come
out again (lnZ and best-fit tables), which pins the device-side logic -- draw order, derived
columns, masks, priors, compaction, top-k -- exactly.  Only the generator itself (Philox in the
draw kernel) is left to statistics: tests/test_gpu_equivalence.py.
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, gold, install_cpu_device_fakes
import torch_pipeline as dp
from triceratops_amd import device_pipeline as prod
from triceratops_amd import funcs, priors

CPU = torch.device("cpu")


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))


def close(a, b, rtol=1e-12):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], b[~fin], equal_nan=True)
    assert np.allclose(a[fin], b[fin], rtol=rtol, atol=1e-300)


def test_samplers_and_relations_match_host_functions():
    g = gold("priors_funcs.npz")
    x = g["uniforms"]
    for M in (1.3, 1.0, 0.7, 0.3, 0.25, 0.1, 0.05):
        close(dp.sample_q(T(x), M), g["sample_q_%g" % M])
        close(dp.sample_q_companion(T(x), M), g["sample_qc_%g" % M])
    close(dp.sample_rp(T(x), T(g["rp_masses"]), False), g["sample_rp"])
    close(dp.sample_rp(T(x), T(g["rp_masses"]), True), g["sample_rp_flat"])
    close(dp.sample_inc(T(x)), g["sample_inc"])
    m = g["sr_masses"]
    r, t = dp.stellar_relations(T(m), T(np.full(m.size, 1.1)), T(np.full(m.size, 6100.0)))
    close(r, g["sr_radii"], 1e-11)
    close(t, g["sr_teffs"], 1e-11)
    for band in ("TESS", "Vis", "J", "H", "K"):
        close(dp.flux_relation(T(m), band), g["flux_relation_" + band], 1e-11)


def test_priors_match_host_functions():
    g = gold("priors_funcs.npz")
    dm = g["prior_dmags"]
    seps, cons = funcs.file_to_contrast_curve(os.path.join(GOLD, "contrast_curve_synth.csv"))
    close(dp._interp(T(dm), cons, seps), np.interp(dm, cons, seps), 1e-13)
    close(dp._interp(T(dm), np.array([1.0]), np.array([2.2])), np.full(dm.size, 2.2))
    for M in (1.25, 0.8):
        for plx in (12.5, np.nan):
            tag = "%g_%s" % (M, "nan" if np.isnan(plx) else "%g" % plx)
            close(dp._bound_rate(M, plx, T(dm), seps, cons, False), g["bound_TP_" + tag], 1e-10)
            close(dp._bound_rate(M, plx, T(dm), seps, cons, True), g["bound_EB_" + tag], 1e-10)
            close(dp._bound_rate(M, plx, T(dm), np.array([2.2]), np.array([1.0]), False),
                  g["bound_TP_nocc_" + tag], 1e-10)


NumpyStreamRng = dp.NumpyStreamRng


G = gold("lnz_cases.npz")
PAR_CASES = [str(c) for c in G["cases"]]      # vector-path and per-draw-loop ("_serial") cases


def _call(name, P, cc, filt, parallel=True):
    s = dict(zip(("M_s", "R_s", "Teff", "Z", "plx", "Tmag", "Jmag", "Hmag", "Kmag"), (float(v) for v in G["star"])))
    base = (G["time"], G["flux"], float(G["sigma"][0]), P, s["M_s"], s["R_s"], s["Teff"])
    tri = os.path.join(GOLD, "trilegal_synth.csv")
    fn = getattr(dp, "lnZ_" + name)
    N = int(G["N"][0]) if parallel else 300
    if name in ("TTP", "TEB"):
        return fn(*base, 0.0, N, parallel)
    if name in ("PTP", "PEB", "STP", "SEB"):
        return fn(*base, 0.0, s["plx"], cc, filt, N, parallel)
    mags = (s["Tmag"], s["Jmag"], s["Hmag"], s["Kmag"])
    if name in ("DTP", "DEB"):
        return fn(*base, 0.0, *mags, tri, cc, filt, N, parallel)
    return fn(*base, *mags, tri, cc, filt, N, parallel)


@pytest.mark.parametrize("case", PAR_CASES)
def test_device_pipeline_reproduces_reference_when_fed_the_numpy_stream(case, monkeypatch):
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd import _lib
    monkeypatch.setattr(_lib, "compute_device", lambda: CPU)
    monkeypatch.setattr(prod, "RNG", NumpyStreamRng())
    name, variant = case.split("_")
    P = [2.5, 4.0] if variant == "range" else 3.3
    cc = os.path.join(GOLD, "contrast_curve_synth.csv") if variant == "ccJ" else None
    np.random.seed(int(G[case + "_seed"][0]))
    res = _call(name, P, cc, "J" if cc else "TESS", parallel=variant != "serial")
    dicts = res if isinstance(res, tuple) else (res,)
    assert len(dicts) == int(G[case + "_nres"][0])
    for i, d in enumerate(dicts):
        want = G["%s_lnZ%d" % (case, i)][0]
        assert (d["lnZ"] == want) if not np.isfinite(want) else abs(d["lnZ"] - want) < 1e-8 + 1e-12 * abs(want), (case, i)
        logw = G["%s_logw%d" % (case, i)]
        # Equal chi^2 values (flat models) have no defined order: with fewer than 100 finite draws
        # compare the finite rows as a set, otherwise the head of the table.
        n_fin = int(np.isfinite(logw).sum())
        for k in ("P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB", "R_EB", "fluxratio_EB",
                  "fluxratio_comp", "M_s", "R_s", "u1", "u2"):
            w = G["%s_res%d_%s" % (case, i, k)]
            assert d[k].shape == (100,)
            if name not in ("TTP", "TEB"):
                continue        # logw = lnL + prior elsewhere: it does not count the finite lnL
            if n_fin <= 100:
                assert np.allclose(np.sort(d[k][:n_fin]), np.sort(w[:n_fin]), rtol=1e-8, atol=0), (case, i, k)
            else:
                assert np.allclose(d[k][:10], w[:10], rtol=1e-8, atol=0), (case, i, k)


def test_sampling_switch(monkeypatch):
    import triceratops_amd
    from triceratops_amd import marginal_likelihoods as ml
    from triceratops_amd import fused
    calls = []
    monkeypatch.setattr(fused, "lnZ_TTP", lambda *a, **k: calls.append(a) or {"lnZ": 0.0})
    triceratops_amd.set_sampling("device")
    try:
        assert isinstance(prod.RNG, prod.TorchRng)
        assert ml.lnZ_TTP(1, 2, 3)["lnZ"] == 0.0 and calls
    finally:
        triceratops_amd.set_sampling("numpy")
    n0 = len(calls)
    triceratops_amd.set_sampling("numpy-device")
    try:
        assert isinstance(prod.RNG, prod.NumpyStreamRng)
        assert ml.lnZ_TTP(1, 2, 3, 4, 5, 6, 7, 8, 100, True)["lnZ"] == 0.0 and len(calls) == n0 + 1
    finally:
        triceratops_amd.set_sampling("numpy")
    with pytest.raises(ValueError):
        triceratops_amd.set_sampling("gpu")
