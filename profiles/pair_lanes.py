"""What the lanes of the pair loop's trips do (instrumented build, profiles/instrumented/pair_lanes_patch.py):
    TRX_LIB=profiles/ab_libs/libtrx_pairlanes.so python profiles/pair_lanes.py [rows] [n_time ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
times = [int(x) for x in sys.argv[2:]] or [100, 2000]
L = _lib.lib()
L.trx_set_skip_excluded(0)
buf = (ctypes.c_ulonglong * 8)()
for n_time in times:
    rng = np.random.default_rng(synth.SEED)
    t_d = _lib.dev(synth.time_grid(n_time))
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    out = torch.empty(n_rows, dtype=torch.float64, device="cuda")
    tot = np.zeros(5)
    for fam in synth.FAMILIES:
        rows = _lib.dev(synth.family_rows(rng, fam, n_rows))
        torch.cuda.synchronize(); L.trx_dbg_lanes(None, 1)
        _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20, out=out)
        torch.cuda.synchronize(); L.trx_dbg_lanes(buf, 0)
        tot += np.array(list(buf)[:5], dtype=float)
    trips, pairs, on, skipped, heavy = tot
    print("n_time %4d: %.3g trips; lanes with a pair %.3f; lanes on the disc %.3f (of the pairs: %.3f); trips that skip the flux stage %.3f; "
          "lanes on the disc in the other trips %.3f; pairs of contact cells %.3f of all pairs"
          % (n_time, trips, pairs / (64 * trips), on / (64 * trips), on / pairs, skipped / trips, on / (64 * (trips - skipped)), heavy / pairs))
