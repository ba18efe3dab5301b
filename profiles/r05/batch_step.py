"""A few 64-TOI steps (BASELINE configs[3] on one GPU) for rocprofv3:   python profiles/r05/batch_step.py [steps] [tois] [N] [streams]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
from triceratops_amd import sharding, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tois = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000
if len(sys.argv) > 4:
    sharding.streams = int(sys.argv[4])
GOLD = os.path.join(ROOT, "tests", "golden")
jobs = synth.toi_jobs(tois, n_time=200, N=N, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
if os.environ.get("NOGC"):
    import gc
    gc.disable()
for s in range(steps + 1):
    torch.manual_seed(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    triceratops_amd.calc_probs_many(jobs)
    torch.cuda.synchronize()
    print("step %d: %.4f s  %s" % (s, time.perf_counter() - t0, {k: round(v, 4) for k, v in sharding.timing.items()}), flush=True)
