"""Model evaluations per cell (trx_set_debug_node_counts) of one mid-transit row on config 1's grid, printed for the first
190 in-window cells: where the chunks of the stencil path meet."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from triceratops_amd import _lib, synth
_lib.require_gpu()
t = np.linspace(-0.0499, 0.0499, 400)   # (dt = 0.18 exposures as in config 1; nearly every cell inside the window)
row = synth.reference_tp_row()
L = _lib.lib()
L.trx_set_debug_node_counts(1)
n = _lib.flux_grid(_lib.MODEL_TP, 0, _lib.dev(t), _lib.dev(row), synth.EXPTIME, synth.NSAMPLES, want_secdepth=False)[0].cpu().numpy()[0]
L.trx_set_debug_node_counts(0)
nz = np.nonzero(n)[0]
print("cells with evaluations:", nz.size, "first", nz[0], "sum", n.sum(), "mean over them %.3f" % n[nz].mean())
seg = n.astype(int)
for k in range(0, seg.size, 58):
    print("%3d" % k, " ".join(str(x) for x in seg[k:k + 58]))
