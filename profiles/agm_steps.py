"""Steps of the shared AGM recurrence (cel_pair) per (cell, node) pair vs the wave's maximum, with an
instrumented build (TRX_LIB=profiles/ab_libs/libtrx_agmcount.so):
    TRX_LIB=... python profiles/agm_steps.py [rows] [n_time ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
times = [int(x) for x in sys.argv[2:]] or [100, 2000]
L = _lib.lib()
L.trx_set_skip_excluded(0)
buf = (ctypes.c_ulonglong * 16)()
for n_time in times:
    rng = np.random.default_rng(synth.SEED)
    t = synth.time_grid(n_time); t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    out = torch.empty(n_rows, dtype=torch.float64, device="cuda")
    for fam in synth.FAMILIES:
        rows = _lib.dev(synth.family_rows(rng, fam, n_rows))
        torch.cuda.synchronize(); L.trx_dbg_read(None, 1)
        _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20, out=out)
        torch.cuda.synchronize(); L.trx_dbg_read(buf, 0)
        v = list(buf)
        hist = np.array(v[3:14], dtype=float)
        print("n_time %4d %-28s lanes' own steps / (64 x wave steps) = %.3f; / (active lanes x wave steps) = %.3f; mean steps %.2f; "
              "steps histogram 1..10+: %s" % (n_time, str(fam[0])[:28], v[0] / max(v[1], 1), v[0] / max(v[2], 1),
                                             v[0] / max(hist.sum(), 1), " ".join("%.2f" % (h / max(hist.sum(), 1)) for h in hist[1:])))
