"""one configuration of profiles/fuzz_bounded.py, both modes, tables side by side:
    python profiles/fuzz_bounded_one.py n_time N kind streams tois toi_seed run_seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import triceratops_amd
from triceratops_amd import _lib, sharding, synth
GOLD = os.path.join(ROOT, "tests", "golden")
n_time, N, kind, streams, n_tois, toi_seed, run_seed = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
wide = kind.endswith("+wide")
kind = kind.replace("+wide", "")
triceratops_amd.set_sampling("device")
sharding.per_unit_seed = True
L = _lib.lib()
L.trx_set_debug_poison(1)
got = {}
for mode in (0, 2, 2):
    L.trx_set_bounded_evaluation(mode)
    np.random.seed(run_seed); torch.manual_seed(run_seed)
    jobs = synth.toi_jobs(n_tois, n_time=n_time, N=N, seed=toi_seed, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                          contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv") if toi_seed & 1 else None)
    r2 = np.random.default_rng(toi_seed + 1)
    for _, kw in jobs:
        t, f, s = kw["time"], kw["flux_0"], kw["flux_err_0"]
        if kind == "noise": kw["flux_0"] = 1.0 + r2.normal(0.0, s, t.size)
        elif kind == "scaled": kw["flux_0"] = 1.0 + (f - 1.0) * float(r2.choice([0.2, 0.5, 3.0]))
        elif kind == "deep": kw["flux_0"] = np.where(np.abs(t) < 0.03, f - 0.2, f)
        elif kind == "shifted": kw["flux_0"] = np.roll(f, t.size // 3)
        elif kind == "quiet": kw["flux_err_0"] = s * 0.1
        if wide: kw["time"] = t * 2.0
    sharding.streams = streams
    out = triceratops_amd.calc_probs_many(jobs)
    if mode in got:
        print("second run of mode 2 equals the first:", all(np.array_equal(a.lnZ, b.lnZ, equal_nan=True) for a, b in zip(out, got[mode])))
    got[mode] = out
for x, z in zip(got[0], got[2]):
    print(x.probs.assign(lnZ0=x.lnZ, lnZ2=z.lnZ)[["scenario", "lnZ0", "lnZ2"]].to_string())
    print("FPP", x.FPP, z.FPP)
for x, z in zip(got[0], got[2]):
    print("lnZ differences:", np.array2string(z.lnZ - x.lnZ, precision=3))
