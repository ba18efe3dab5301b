"""Wave cycles by phase of cells_kernel, one row per wave vs batches of rows (debug build -DTRX_PHASE_TIMERS):
TRX_LIB=profiles/ab_libs/libtrx_dbg.so python profiles/phase_cycles2.py <n_time> <rows>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
rng = np.random.default_rng(1)
t = synth.time_grid(n_time); t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
L = _lib.lib()
L.trx_set_skip_excluded(0)      # throughput of the model: every row counted is evaluated
L.trx_debug_phase_cycles.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 8)()
names = {0: ["row blocks", "window pass", "plans", "pairs", "-", "rest", "-", "total"],
         1 << 30: ["row blocks", "window pass", "plans", "pairs", "-", "rest", "-", "total"]}
for fam in (synth.FAMILIES[0], synth.FAMILIES[1]):
    rows = _lib.dev(synth.family_rows(rng, fam, n))
    for below in (0, 1 << 30):
        L.trx_set_cell_packing_below(below)
        _lib.lnl_batch(fam[1], 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20); torch.cuda.synchronize()
        L.trx_debug_phase_cycles(out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _lib.lnl_batch(fam[1], 0, t_d, f_d, synth.SIGMA, rows, synth.EXPTIME, 20); b.record(); torch.cuda.synchronize()
        L.trx_debug_phase_cycles(out)
        v = np.array(list(out), dtype=float)
        print("%s n_time=%d %s: %.3f ms; wave-cycles per row %.0f;" % (fam[0], n_time, "batches" if below else "one row", a.elapsed_time(b), v[7] / n),
              " ".join("%s %.1f%%" % (nm, 100 * x / v[7]) for nm, x in zip(names[below], v) if nm not in ("-", "total")))
L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW)
