"""Device-mode calc_probs on the inputs of the reference's example notebooks, 20 seeds each at
N = 1e6, against the numbers those notebooks printed (the only reference results that passed
through the real pytransit).  Writes the tables tests/test_gpu_notebook_anchors.py asserts on.
    python profiles/notebook_anchors.py [n_seeds] [sampling] > profiles/r03/notebook_anchors.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import anchors  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sampling = sys.argv[2] if len(sys.argv) > 2 else "device"
raw_out = sys.argv[3] if len(sys.argv) > 3 else None      # .npz with every run's lnZ / prob / FPP / R_p
raw = {}
np.set_printoptions(linewidth=200, precision=4)
for case in anchors.CASES:
    t0 = time.perf_counter()
    lnZ, prob, fpp, rp = anchors.run_many(case, range(1000, 1000 + n_seeds), sampling=sampling)
    dt = time.perf_counter() - t0
    raw.update({case + "_lnZ": lnZ, case + "_prob": prob, case + "_FPP": fpp, case + "_Rp": rp})
    nb_prob, nb_fpp, nb_rp = anchors.notebook(case)
    print("==== %s: %d runs, N = 1e6, %s sampling, %.2f s per run" % (case, n_seeds, sampling, dt / n_seeds))
    print("%-7s %12s %12s %12s %12s   %12s" % ("", "mean lnZ", "std lnZ", "mean prob", "std prob", "notebook prob"))
    for j, s in enumerate(anchors.SCENARIOS):
        fin = np.isfinite(lnZ[:, j])
        m = lnZ[fin, j].mean() if fin.any() else -np.inf
        sd = lnZ[fin, j].std() if fin.any() else 0.0
        print("%-7s %12.3f %12.3f %12.4e %12.4e   %12.4e" % (s, m, sd, prob[:, j].mean(), prob[:, j].std(), nb_prob[j]))
    sh = anchors.free_shares(prob)
    nb_sh = anchors.free_shares(nb_prob)[0]
    lg = np.log(sh)
    print("TRILEGAL-free shares TP:PTP:STP  ours mean", sh.mean(0), " notebook", nb_sh)
    print("  log-share mean", lg.mean(0), "std", lg.std(0, ddof=1), " notebook", np.log(nb_sh),
          " z", (np.log(nb_sh) - lg.mean(0)) / lg.std(0, ddof=1))
    print("FPP: ours mean %.5f std %.5f median %.5f  (runs: %s)" % (fpp.mean(), fpp.std(), np.median(fpp),
                                                                    np.array2string(np.sort(fpp), precision=4)))
    print("FPP notebook single run %.4g" % nb_fpp)
    if case == "toi465_nocc":
        print("FPP notebook 20 runs: %.4f +- %.4f" % tuple(anchors.A["toi465_FPP20_nocc"]))
    if case == "toi465_cc":
        print("FPP notebook 20 runs: %.4f +- %.4f" % tuple(anchors.A["toi465_FPP20_cc"]))
    print("best-fit TP R_p: ours mean %.3f std %.3f  notebook %.3f" % (rp.mean(), rp.std(), nb_rp))
    sys.stdout.flush()
if raw_out:
    np.savez_compressed(raw_out, **raw)
