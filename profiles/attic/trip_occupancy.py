"""Occupancy of the pair loop's 64-lane trips in the batched variant: (cell, node) pairs of a batch of B rows
over 64 x ceil(pairs / 64), from the kernel's own node counts (trx_set_debug_node_counts):
    python profiles/trip_occupancy.py [rows] [n_time ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
times = [int(x) for x in sys.argv[2:]] or [100, 200]
L = _lib.lib()
for n_time in times:
    rng = np.random.default_rng(synth.SEED)
    t_d = _lib.dev(synth.time_grid(n_time))
    B = max(1, min(22, (640 + n_time // 2) // n_time))
    tot_pairs = tot_slots = tot_cells = 0
    hist = np.zeros(12)
    for fam in synth.FAMILIES:
        rows = _lib.dev(synth.family_rows(rng, fam, n_rows))
        L.trx_set_debug_node_counts(1)
        try:
            c, _ = _lib.flux_grid(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, rows, synth.EXPTIME, 20, False)
        finally:
            L.trx_set_debug_node_counts(0)
        per_row = c.sum(dim=1).cpu().numpy()
        nb = per_row.size // B
        per_batch = per_row[:nb * B].reshape(nb, B).sum(axis=1)
        trips = np.ceil(per_batch / 64)
        tot_pairs += per_batch.sum(); tot_slots += 64 * trips.sum(); tot_cells += nb * B * n_time
        hist += np.bincount(np.minimum(trips.astype(int), 11), minlength=12)
    print("n_time %4d, B = %d: %.2f pairs per cell, %.1f pairs per batch, trips per batch %.2f, lane occupancy of the trips %.3f"
          % (n_time, B, tot_pairs / tot_cells, tot_pairs / (tot_cells / (B * n_time)), tot_slots / 64 / (tot_cells / (B * n_time)),
             tot_pairs / tot_slots))
    print("     share of batches by trips 0..11+: " + " ".join("%.3f" % (h / hist.sum()) for h in hist))
