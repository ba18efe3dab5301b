"""Mixed-precision (TRX_FLAG_FP32_MODEL) vs fp64 model: error statistics and speed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from triceratops_amd import _lib, synth
rng = np.random.default_rng(2)
for nt in (2000, 200):
    t = synth.time_grid(nt); t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
    for name, model, rows in (("TP", 0, synth.tp_rows(rng, 20000, True)), ("EB", 1, synth.eb_rows(rng, 20000, False, True)),
                              ("EBx2P", 2, synth.eb_rows(rng, 20000, True, True))):
        r_d = _lib.dev(rows)
        g64, _ = _lib.flux_grid(model, 0, t_d, r_d[:, :2000].contiguous(), synth.EXPTIME, 20, False)
        g32, _ = _lib.flux_grid(model, _lib.FLAG_FP32_MODEL, t_d, r_d[:, :2000].contiguous(), synth.EXPTIME, 20, False)
        dflux = (g32 - g64).abs()
        h64 = _lib.lnl_batch(model, 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20)
        h32 = _lib.lnl_batch(model, _lib.FLAG_FP32_MODEL, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20)
        fin = torch.isfinite(h64)
        dh = (h32 - h64)[fin].abs()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): _lib.lnl_batch(model, 0, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20)
        torch.cuda.synchronize(); t64 = (time.perf_counter() - t0) / 3; t0 = time.perf_counter()
        for _ in range(3): _lib.lnl_batch(model, _lib.FLAG_FP32_MODEL, t_d, f_d, synth.SIGMA, r_d, synth.EXPTIME, 20)
        torch.cuda.synchronize(); t32 = (time.perf_counter() - t0) / 3
        print("n_time %4d %-6s flux |d| max %.2e mean %.2e | chi2/2 |d| max %.3g median %.3g rel-max %.2e | fp64 %.2f ms fp32 %.2f ms (x%.2f) %.3e evals/s"
              % (nt, name, dflux.max(), dflux.mean(), dh.max(), dh.median(), (dh / h64[fin]).max(), t64 * 1e3, t32 * 1e3, t64 / t32, 20000 * nt / t32))
