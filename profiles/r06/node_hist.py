"""Evaluations per cell of config 1's EB rows (TRX_FLAG_COUNT_EVALUATIONS), saved for comparison between builds:
    TRX_LIB=... python profiles/r06/node_hist.py out.npy"""
import sys
import numpy as np
sys.path.insert(0, ".")
from triceratops_amd import _lib, synth
rng = np.random.default_rng(1)
t = synth.time_grid(2000)
rows = synth.eb_rows(rng, 2000)
n = _lib.flux_grid(_lib.MODEL_EB, _lib.FLAG_COUNT_EVALUATIONS, _lib.dev(t), _lib.dev(rows), synth.EXPTIME, 20, False)[0].cpu().numpy()
f = _lib.flux_grid(_lib.MODEL_EB, 0, _lib.dev(t), _lib.dev(rows), synth.EXPTIME, 20, False)[0].cpu().numpy()
np.save(sys.argv[1], n.astype(np.int8))
np.save(sys.argv[1] + ".flux.npy", f)
u, c = np.unique(n, return_counts=True)
print("evaluations per cell %.4f;" % n.mean(), " ".join("%d:%.1f" % (a, b / 2000) for a, b in zip(u, c)), "(cells per row)")
