import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
rng = np.random.default_rng(synth.SEED)
t = synth.time_grid(100); t_d = _lib.dev(t)
f_d = _lib.dev(1.0 + rng.normal(0, synth.SIGMA, 100))
for fam in synth.FAMILIES:
    rows = _lib.dev(synth.family_rows(rng, fam, 20000))
    h = _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
    print(fam[0], fam[1], "excluded fraction %.3f" % float(torch.isinf(h).double().mean()))
