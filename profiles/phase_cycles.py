"""Where the wave cycles of rows_kernel go (debug build with -DTRX_PHASE_TIMERS, see profiles/phase_cycles.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(1)
t = synth.time_grid(n_time); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
L = _lib.lib()
out = (ctypes.c_ulonglong * 8)()
L.trx_debug_phase_cycles(out)
names = ["prologue (phases 1-3)", "cell plans", "stage A (orbit nodes, filing)", "stage B (Mandel-Agol)", "stage C + rest of the time loop"]
for fam in (synth.FAMILIES[0], synth.FAMILIES[1], synth.FAMILIES[2]):
    rows = _lib.dev(synth.family_rows(rng, fam, n))
    _lib.lnl_batch(fam[1], 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
    L.trx_debug_phase_cycles(out)
    v = np.array(list(out)[:5], dtype=float)
    print(fam[0], " ".join("%s %.1f%%" % (nm, 100 * x / v.sum()) for nm, x in zip(names, v)))
