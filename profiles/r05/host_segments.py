"""Host time of a 64-target calc_probs_many by segment (perf_counter around a handful of functions, not cProfile: its
per-call overhead doubles the small ones)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import triceratops_amd  # noqa: E402
from triceratops_amd import _lib, device_pipeline as dp, fused, sharding, synth  # noqa: E402
from triceratops_amd import marginal_likelihoods as ml  # noqa: E402

acc = {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc.setdefault(label, [0.0, 0])
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(obj, name, g)


wrap(fused._Scenario, "__init__", "Scenario.__init__")
wrap(fused._Scenario, "field", "Scenario.field")
wrap(fused._Scenario, "run", "Scenario.run")
wrap(fused._Scenario, "_run_native", "Scenario._run_native")
wrap(fused, "_on_device")
wrap(_lib, "dev", "_lib.dev")
wrap(dp._Field, "__init__", "dp._Field.__init__")
wrap(dp._Field, "need_ldc", "dp._Field.need_ldc")
wrap(ml._Field, "__init__", "ml._Field.__init__ (host)")
wrap(fused, "flush")
wrap(fused, "records_to_rows")
GOLD = os.path.join(ROOT, "tests", "golden")
jobs = synth.toi_jobs(64, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
for s in range(4):
    torch.manual_seed(s)
    acc.clear()
    t0 = time.perf_counter()
    triceratops_amd.calc_probs_many(jobs)
    dt = time.perf_counter() - t0
print("step %.4f s  %s" % (dt, {k: round(v, 4) for k, v in sharding.timing.items()}))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-28s %5d calls %7.2f ms  (%.1f us each)" % (k, n, 1e3 * t, 1e6 * t / n))
