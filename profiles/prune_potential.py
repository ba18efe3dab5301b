"""How many masked draws of a real lnZ_* call could be dropped by a bound on their chi^2?  For TOI-465.01
(S/N ~ 70) and TOI-411.02 (S/N ~ 10): distribution of h - h_min over the masked draws of lnZ_TTP / lnZ_TEB,
and the share of draws whose chi^2/2 over a PROBE of k points (every (n/k)-th time stamp) plus the
out-of-window points already exceeds h_min + 90.    python profiles/prune_potential.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import anchors
import triceratops_amd
from triceratops_amd import fused, _lib

triceratops_amd.set_sampling("device")
for case in ("toi465_nocc", "toi411", "kep10"):
    stars, t, f, sigma, P = anchors.inputs(case)
    M_s, R_s, Teff = (float(stars[c][0]) for c in ("mass", "rad", "Teff"))
    for name in ("lnZ_TTP", "lnZ_TEB"):
        fused.DUMP = []
        torch.manual_seed(1)
        getattr(fused, name)(t, f, sigma, P, M_s, R_s, Teff, 0.0, 1_000_000, True, anchors.CASES[case]["mission"])
        d = fused.DUMP[0]; fused.DUMP = None
        planet = name == "lnZ_TTP"
        idx = torch.nonzero(d["mask"]).flatten()
        nblk = 10 if planet else 11
        block = d["cols"][:nblk].index_select(1, idx).contiguous()
        model = _lib.MODEL_TP if planet else _lib.MODEL_EB
        t_d, f_d = _lib.dev(t), _lib.dev(f)
        L = _lib.lib(); L.trx_set_skip_excluded(0)
        grid, _ = _lib.flux_grid(model, 0, t_d, block, 0.00139, 20, want_secdepth=False)
        L.trx_set_skip_excluded(1)
        h = _lib.lnl_batch(model, 0, t_d, f_d, sigma, block, 0.00139, 20)
        fin = torch.isfinite(h)
        term = ((f_d[None, :] - grid) ** 2) / (2 * sigma ** 2)          # per-cell chi^2/2
        hmin = h[fin].min()
        dh = (h[fin] - hmin).cpu().numpy()
        print("%s %s: %d masked draws (%d finite), n_time %d, h_min %.1f; share with h - h_min > 90: %.3f; > 40: %.3f"
              % (case, name, idx.numel(), int(fin.sum()), t.size, float(hmin), (dh > 90).mean(), (dh > 40).mean()))
        inw = grid != 1.0                                              # cells with a model value other than 1
        out_part = (term * (~inw)).sum(1)
        for k in (0, 4, 8, 16, 32):
            probe = torch.zeros(t.size, dtype=torch.bool, device=grid.device)
            if k: probe[torch.linspace(0, t.size - 1, k).round().long()] = True
            lb = out_part + (term * (inw & probe[None, :])).sum(1)
            frac = ((lb[fin] - hmin) > 90).float().mean().item()
            cells_left = (inw[fin] & ~probe[None, :])[(lb[fin] - hmin) <= 90].sum().item() + (inw[fin] & probe[None, :]).sum().item()
            print("    probe %2d points: %.3f of the draws dropped; model evaluations left %.3f of %d"
                  % (k, frac, cells_left / max(1, inw[fin].sum().item()), inw[fin].sum().item()))
