"""one pass of the 18 families through the likelihood kernel at a given n_time (for rocprofv3 --pmc):
python profiles/cells_once.py <n_time> <rows> [cells|rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
which = sys.argv[3] if len(sys.argv) > 3 else "cells"
L = _lib.lib()
L.trx_set_skip_excluded(0)      # throughput of the model: every row counted is evaluated
L.trx_set_cell_packing_below((1 << 30) if which == "cells" else 0)
rng = np.random.default_rng(synth.SEED)
t = synth.time_grid(n_time); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
out = torch.empty(n_rows, dtype=torch.float64, device="cuda")
for fam in synth.FAMILIES:
    r_d = _lib.dev(synth.family_rows(rng, fam, n_rows))
    _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20, out=out)
torch.cuda.synchronize()
