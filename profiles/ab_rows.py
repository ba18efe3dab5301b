"""A/B of rows-per-wave (B) on the fused kernel: python profiles/ab_rows.py [n_rows] [n_time]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(1)
t = synth.time_grid(nt); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
blocks = [(0, _lib.dev(synth.tp_rows(rng, n, True))), (1, _lib.dev(synth.eb_rows(rng, n, False, True)))]
L = _lib.lib()
for B in (0, 1, 2, 4, 8, 16):
    L.trx_set_rows_per_wave(B)
    for model, rows in blocks:
        _lib.lnl_batch(model, 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        for model, rows in blocks:
            _lib.lnl_batch(model, 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print("B=%2d  %.3f ms/launch  %.3e evals/s" % (B, dt * 1e3, n * nt / dt))
L.trx_set_rows_per_wave(0)
