"""native scenario call vs the torch-operator path on the bench's batch jobs: python profiles/native_check.py [tois] [N] [threads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import triceratops_amd
from triceratops_amd import fused, sharding, synth, _lib
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
tois = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
jobs = synth.toi_jobs(tois, n_time=200, N=N, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
triceratops_amd.set_threads(thr)
sharding.per_unit_seed = True
res = {}
for native in (True, False, True):
    fused.NATIVE = native
    np.random.seed(100); torch.manual_seed(100)
    out = triceratops_amd.calc_probs_many(jobs)
    res.setdefault(native, []).append(np.array([tg.lnZ for tg in out]))
    print("native" if native else "torch ", "FPP", ["%.6f" % tg.FPP for tg in out])
a, b, c = res[True][0], res[False][0], res[True][1]
print("native == native again:", np.array_equal(a, c, equal_nan=True), " native == torch:", np.array_equal(a, b, equal_nan=True))
bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
for t, k in bad[:20]:
    print("  TOI %d scenario %d: native %.17g torch %.17g" % (t, k, a[t, k], b[t, k]))
