"""Stage-A trip count per 64-cell chunk = max node count over its lanes: how much of it is the tail
run for the few all-sub-exposure cells?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
rng = np.random.default_rng(3)
t_d = _lib.dev(synth.time_grid(2000))
L = _lib.lib()
tot = np.zeros(4)
for fam in synth.FAMILIES[:6]:
    rows = _lib.dev(synth.family_rows(rng, fam, 1024))
    L.trx_set_debug_node_counts(1)
    try:
        g, _ = _lib.flux_grid(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, rows, synth.EXPTIME, 20, False)
    finally:
        L.trx_set_debug_node_counts(0)
    n = torch.nn.functional.pad(g, (0, 48)).reshape(g.shape[0], 32, 64)
    trips = n.max(dim=2).values                                  # stage-A iterations per chunk
    tier = torch.where(n == 20, torch.zeros_like(n), n)
    trips_tier = tier.max(dim=2).values
    nfull = (n == 20).sum(dim=2)
    lane_work = n.sum(dim=2)                                     # lane-iterations actually needed
    print("%-8s trips/chunk %.2f  (tier cells only %.2f)  lane-iterations/chunk %.1f  chunks with full cells %.3f, full cells in them %.2f (max %d)"
          % (fam[0], trips.mean(), trips_tier.mean(), lane_work.mean(), (nfull > 0).double().mean(),
             nfull[nfull > 0].double().mean(), nfull.max()))
