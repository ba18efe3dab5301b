"""rows per wave (B) sweep of the packed-cell kernel: python profiles/cells_batch_sweep.py [rows] [n_time ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
times = [int(x) for x in sys.argv[2:]] or [100, 200]
L = _lib.lib()
L.trx_set_skip_excluded(0)      # throughput of the model: every row counted is evaluated
L.trx_set_cell_packing_below(1 << 30)
for n_time in times:
    rng = np.random.default_rng(synth.SEED)
    t = synth.time_grid(n_time); t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    blocks = [(_lib.dev(synth.family_rows(rng, fam, n_rows)), fam) for fam in synth.FAMILIES]
    out = torch.empty(n_rows, dtype=torch.float64, device="cuda")
    line = "n_time %4d, %d rows:" % (n_time, n_rows)
    for B in (0, 1, 2, 3, 4, 6, 8, 11, 16, 22):
        L.trx_set_rows_per_wave(B)
        def step():
            for r_d, fam in blocks:
                _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, r_d,
                               synth.EXPTIME, 20, out=out)
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): step()
        b.record(); torch.cuda.synchronize()
        line += "  B=%s %.2f" % (B if B else "auto", a.elapsed_time(b) / 3)
    print(line + "  (ms per 18 launches)")
L.trx_set_rows_per_wave(0)
L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW)
