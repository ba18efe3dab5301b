"""Wave cycles by phase of the bounded evaluation's passes over a 16-TOI step (TRX_LIB = a -DTRX_PHASE_TIMERS build):
set-up of a batch (0), window pass (1), cell plans (2), pair table + pair trips (3), finalisation (5), everything (7)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import triceratops_amd  # noqa: E402
from triceratops_amd import _lib, sharding, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
L = _lib.lib()
L.trx_debug_phase_cycles.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 8)()
jobs = synth.toi_jobs(16, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
sharding.streams = 1
for s in range(2):
    torch.manual_seed(s)
    triceratops_amd.calc_probs_many(jobs)
    torch.cuda.synchronize()
    L.trx_debug_phase_cycles(out)
v = list(out)
tot = v[7]
names = {0: "batch set-up", 1: "window pass", 2: "cell plans", 3: "pair table + trips", 5: "finalisation"}
print("wave cycles of cells_kernel over a 16-TOI step: %.3e" % tot)
acc = 0
for k, n in names.items():
    print("  %-20s %5.1f %%" % (n, 100.0 * v[k] / tot))
    acc += v[k]
print("  %-20s %5.1f %%" % ("other (loops, verdicts, lists)", 100.0 * (tot - acc) / tot))
