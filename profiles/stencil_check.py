"""Centre-value stencil (dense uniform time grids) against the Gauss-node path of the same kernel and
against every sub-exposure evaluated: max |dflux| over the 18 bench families, share of the cells that
take the stencil, evaluations per cell, launch time.   python profiles/stencil_check.py [n_time] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
L = _lib.lib()
L.trx_set_skip_excluded(0)      # throughput of the model: every row counted is evaluated
rng = np.random.default_rng(synth.SEED)
t = synth.time_grid(n_time); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
worst = worst_all = 0.0
ev_on = ev_off = cells = st_cells = 0.0
for fam in synth.FAMILIES:
    rows = _lib.dev(synth.family_rows(rng, fam, nr))
    fl = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
    res = {}
    for name, st, tiers in (("on", 1, 1), ("off", 0, 1), ("all", 0, 0)):
        L.trx_set_stencil(st); L.trx_set_supersample_tiers(tiers)
        res[name] = _lib.flux_grid(fam[1], fl, t_d, rows, synth.EXPTIME, 20, False)[0]
        if name != "all":
            L.trx_set_debug_node_counts(1)
            res["n" + name] = _lib.flux_grid(fam[1], fl, t_d, rows, synth.EXPTIME, 20, False)[0]
            L.trx_set_debug_node_counts(0)
    L.trx_set_stencil(1); L.trx_set_supersample_tiers(1)
    d = float((res["on"] - res["off"]).abs().max()); da = float((res["on"] - res["all"]).abs().max())
    worst, worst_all = max(worst, d), max(worst_all, da)
    ev_on += float(res["non"].sum()); ev_off += float(res["noff"].sum()); cells += res["on"].numel()
    st_cells += float(((res["non"] == 1) & (res["noff"] > 1)).sum())
    print("%-8s max|stencil - gauss| = %.2e   max|stencil - all sub-exposures| = %.2e   evals/cell %.3f -> %.3f" % (
        fam[0], d, da, float(res["noff"].mean()), float(res["non"].mean())))
print("n_time %d: worst %.2e (vs all sub-exposures %.2e); evaluations per cell %.3f -> %.3f; %.1f %% of the cells take the stencil"
      % (n_time, worst, worst_all, ev_off / cells, ev_on / cells, 100 * st_cells / cells))
# timing, likelihood path
blocks = [(_lib.dev(synth.family_rows(rng, fam, 100000 if n_time <= 2000 else 20000)), fam) for fam in synth.FAMILIES]
out = torch.empty(blocks[0][0].shape[1], dtype=torch.float64, device="cuda")
hs = {}
for name, st in (("stencil", 1), ("gauss", 0)):
    L.trx_set_stencil(st)
    def step():
        for r_d, fam in blocks:
            _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20, out=out)
    step(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): step()
    b.record(); torch.cuda.synchronize()
    hs[name] = out.clone()
    print("%s: %.2f ms per 18 launches = %.3g evals/s" % (name, a.elapsed_time(b) / 3, n_time * out.numel() * 18 / (a.elapsed_time(b) / 3) * 1e3))
L.trx_set_stencil(1)
fin = torch.isfinite(hs["gauss"])
print("chi2/2 of the last family: max relative difference %.2e" % float(((hs["stencil"][fin] - hs["gauss"][fin]).abs() / hs["gauss"][fin].abs().clamp_min(1e-300)).max()))
