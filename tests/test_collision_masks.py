"""Collision masks of the eclipsing-binary branches (reference tests/test_beb_collision_mask.py):
the q < 0.95 branch at P_orb must be cut by the standard-period contact test, the q >= 0.95 twin
branch at 2 P_orb by the twin-period one (marginal_likelihoods.py:2298-2316, 2716-2740) -- checked
here by behaviour, on geometries where exactly one of the two tests fires."""
import numpy as np
import pytest

from helpers import gold, install_cpu_device_fakes

RSUN, G_, MSUN = 6.957e10, 6.6743e-8, 1.988409870698051e33


def _sma(M, P):
    return ((G_ * M * MSUN) / (4 * np.pi ** 2) * (P * 86400) ** 2) ** (1 / 3)


def _branches(path, radius_eb, P_orb, N=64):
    g = gold("lnz_cases.npz")
    rng = np.random.default_rng(3)
    qs = np.where(np.arange(N) % 2 == 0, 0.5, 0.97)
    masses = qs * 1.0
    z = np.zeros(N)
    cols = (np.full(N, P_orb), qs, np.full(N, 90.0), z, rng.uniform(0, 360, N), masses,
            np.full(N, radius_eb), np.full(N, 1e-6))
    tail = (1.0, 1.0, 0.4, 0.25, None, False, None, None)
    if path == "host":
        from triceratops_amd import marginal_likelihoods as ml
        return ml._binary_branches(g["time"], g["flux"], float(g["sigma"][0]), N, True, 0.00139, 20,
                                   *cols, *tail)
    import torch
    import torch_pipeline as dp
    ctx = dp._ctx(g["time"], g["flux"], float(g["sigma"][0]), N, 0.00139, 20)
    return dp._binary_branches(ctx, *(torch.as_tensor(c) for c in cols), *tail)


@pytest.fixture(params=["host", "device"])
def path(request, monkeypatch):
    """host-sampling and device-resident pipelines, both on the CPU stand-in"""
    import torch
    from triceratops_amd import _lib
    install_cpu_device_fakes(monkeypatch)
    monkeypatch.setattr(_lib, "compute_device", lambda: torch.device("cpu"))
    return request.param


def test_standard_branch_is_cut_by_the_standard_period_contact(path):
    """R_EB + R_s > a(P) but 2 R_s < a(2P): only the twin branch survives"""
    P = 0.3
    a, a2 = _sma(1.5, P) / RSUN, _sma(1.5, 2 * P) / RSUN
    assert 1.0 + 1.4 > a and 2.0 < a2
    res, twin = _branches(path, 1.4, P)
    assert res["lnZ"] == -np.inf and np.isfinite(twin["lnZ"])
    assert np.all(twin["P_orb"][:10] == 2 * P) and np.all(twin["M_EB"][:10] == 0.97)


def test_twin_branch_is_cut_by_the_twin_period_contact(path):
    """R_EB + R_s < a(P) but 2 R_s > a(2P) (tiny companion, very short period): only the
    standard branch survives"""
    P = 0.105
    a, a2 = _sma(1.5, P) / RSUN, _sma(1.97, 2 * P) / RSUN
    assert 1.0 + 0.02 < a and 2.0 > a2
    res, twin = _branches(path, 0.02, P)
    assert np.isfinite(res["lnZ"]) and twin["lnZ"] == -np.inf
    assert np.all(res["M_EB"][:10] == 0.5)


def test_no_contact_keeps_both_branches(path):
    res, twin = _branches(path, 0.5, 3.3)
    assert np.isfinite(res["lnZ"]) and np.isfinite(twin["lnZ"])
