"""How many model evaluations the cells of the bench workload take (trx_set_debug_node_counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
rng = np.random.default_rng(3)
t_d = _lib.dev(synth.time_grid(2000))
L = _lib.lib()
tot = {}
for fam in synth.FAMILIES:
    rows = _lib.dev(synth.family_rows(rng, fam, 2048))
    L.trx_set_debug_node_counts(1)
    try:
        g, _ = _lib.flux_grid(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, rows, synth.EXPTIME, 20, False)
    finally:
        L.trx_set_debug_node_counts(0)
    f, _ = _lib.flux_grid(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, rows, synth.EXPTIME, 20, False)
    vals, cnt = torch.unique(g, return_counts=True)
    frac = {int(v): c / g.numel() for v, c in zip(vals.tolist(), cnt.tolist())}
    for k, v in frac.items():
        tot[k] = tot.get(k, 0) + v / len(synth.FAMILIES)
    occ = (f < 1).double().mean().item()
    print("%-8s" % fam[0], " ".join("n=%d: %.3f" % kv for kv in sorted(frac.items())), " mean evals/cell %.2f  occulted cells %.3f" % (g.mean().item(), occ))
print("all     ", " ".join("n=%d: %.3f" % kv for kv in sorted(tot.items())), " mean evals/cell %.2f" % sum(k * v for k, v in tot.items()))
