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
