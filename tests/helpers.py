"""Shared test helpers: golden loaders and an oracle-backed stand-in for the device calls, so the
host logic (sampling order, masks, priors, tables) can be checked on a CPU-only box.  The
stand-in is test infrastructure: the product has no CPU path."""
import os

import numpy as np
import torch

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def install_cpu_device_fakes(monkeypatch):
    """route triceratops_amd._lib's device entry points to the CPU oracle (tests only)"""
    from triceratops_amd import _lib

    def dev(x, device=None):
        if isinstance(x, torch.Tensor):
            return x.to(dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64))

    def lnl_batch(model, flags, t, f, sigma, params, exptime, nsamples, out=None):
        h = O.lnl_batch(model, t.numpy(), f.numpy(), sigma, params.numpy().reshape(params.shape),
                        companion_is_host=bool(flags & 1), exptime=exptime, nsamples=nsamples,
                        scalar_k=bool(flags & 2)) if params.shape[1] else np.empty(0)
        return torch.as_tensor(h)

    def lnz_scenario(model, flags, t, f, sigma, params, exptime, nsamples, lnprior, n_total, lnsigma):
        h = lnl_batch(model, flags, t, f, sigma, params, exptime, nsamples).numpy()
        logw = np.full(n_total, -np.inf)
        lw = -0.5 * np.log(2 * np.pi) - lnsigma - h
        if lnprior is not None:
            lw = lw + lnprior.numpy()
        logw[:h.size] = lw
        return torch.as_tensor(h), torch.tensor([O.log_mean_exp(logw, n_total)], dtype=torch.float64)

    def flux_grid(model, flags, t, params, exptime, nsamples, want_secdepth=True):
        if model == _lib.MODEL_RAW:
            p = params.numpy()
            g = O.evaluate_pv(t.numpy(), p[:7].T, p[7:].T, exptime, nsamples)
            return torch.as_tensor(g), None
        g, s = O.flux_grid(model, t.numpy(), params.numpy(), companion_is_host=bool(flags & 1),
                           exptime=exptime, nsamples=nsamples, scalar_k=bool(flags & 2))
        return torch.as_tensor(g), torch.as_tensor(s)

    def log_mean_exp(x, n_total):
        return torch.tensor([O.log_mean_exp(x.numpy(), n_total)], dtype=torch.float64)

    monkeypatch.setattr(_lib, "dev", dev)
    monkeypatch.setattr(_lib, "lnl_batch", lnl_batch)
    monkeypatch.setattr(_lib, "lnz_scenario", lnz_scenario)
    monkeypatch.setattr(_lib, "flux_grid", flux_grid)
    monkeypatch.setattr(_lib, "log_mean_exp", log_mean_exp)
    monkeypatch.setattr(_lib, "require_gpu", lambda: None)


def call_extra(ml, case, g):
    """invoke one of the four lnZ_* that calc_probs never calls, as make_golden.py did"""
    tri = os.path.join(GOLD, "trilegal_synth.csv")
    b = (g["time"], g["flux"], float(g["sigma"][0]))
    name, variant = case.split("_")
    par = variant != "serial"
    N = 2000 if par else 300
    if name == "NTPu":
        return ml.lnZ_NTP_unknown(*b, 3.3, 30.0 if variant == "empty" else 14.0, tri, N, par)
    if name == "NEBu":
        return ml.lnZ_NEB_unknown(*b, 3.3, 30.0 if variant == "empty" else 14.0, tri, N, par)
    if name == "NTPe":
        return ml.lnZ_NTP_evolved(*b, 3.3, 3.2, 4900.0, 0.0, N, par)
    return ml.lnZ_NEB_evolved(*b, [3.0, 3.6] if par else 3.3, 3.2, 4900.0, 0.0, N, par)


def check_extra(res, case, g, lnz_tol):
    dicts = res if isinstance(res, tuple) else (res,)
    assert len(dicts) == int(g[case + "_nres"][0])
    for i, d in enumerate(dicts):
        assert sorted(d.keys()) == [str(k) for k in g["%s_keys%d" % (case, i)]], (case, i)
        want = g["%s_lnZ%d" % (case, i)][0]
        assert (d["lnZ"] == want) if not np.isfinite(want) else abs(d["lnZ"] - want) < lnz_tol
        key = "%s_logw%d" % (case, i)
        n_fin = min(100, int(np.isfinite(g[key]).sum())) if key in g.files else 1
        for k in d:
            if k == "lnZ":
                continue
            w = g["%s_res%d_%s" % (case, i, k)]
            v = np.atleast_1d(np.asarray(d[k], dtype=float))
            assert v.shape == w.shape
            top = n_fin if v.size > 1 else 1
            assert np.allclose(v[:top], w[:top], rtol=1e-12, atol=0), (case, i, k)
