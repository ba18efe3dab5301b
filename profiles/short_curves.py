"""Throughput of the likelihood kernel on SHORT light curves (the reference's operating point:
100-200 binned points), 18 scenario families: cells_kernel with one row per wave (LONG) and with a
batch of rows per wave.
usage: python profiles/short_curves.py [rows_per_family [n_time ...]]   (TRX_LIB selects an A/B build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth

JITTER = "--jitter" in sys.argv          # irregular stamps (+-0.3 of the spacing): no centre-value stencil
if JITTER:
    sys.argv.remove("--jitter")
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = _lib.lib()
L.trx_set_skip_excluded(0)      # throughput of the model: every row counted is evaluated
print("# %s, %d rows per family, 18 families; evals/s = n_time x rows x 18 / time of the 18 launches" % (
    os.path.basename(_lib.LIB_PATH), n_rows))
for n_time in ([int(x) for x in sys.argv[2:]] or (50, 100, 150, 200, 250, 300, 400, 500, 1000, 2000)):
    rng = np.random.default_rng(synth.SEED)
    t = synth.time_grid(n_time)
    if JITTER:
        t = np.sort(t + rng.uniform(-0.3, 0.3, n_time) * (t[1] - t[0]))
    t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    nr = n_rows if n_time <= 500 else n_rows // 4
    blocks = [(_lib.dev(synth.family_rows(rng, fam, nr)), fam) for fam in synth.FAMILIES]
    out = torch.empty(nr, dtype=torch.float64, device="cuda")
    line = "n_time %5d:" % n_time
    for name, below in (("one row per wave", 0), ("row batches", 1 << 30)):
        L.trx_set_cell_packing_below(below)
        def step():
            for r_d, fam in blocks:
                _lib.lnl_batch(fam[1], _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0, t_d, f_d, synth.SIGMA, r_d,
                               synth.EXPTIME, 20, out=out)
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): step()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        line += "  %s %.2f ms = %.3g/s" % (name, ms, n_time * nr * 18 / ms * 1e3)
    L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW)
    print(line)
