"""Where the host time of a 64-target calc_probs_many goes (cProfile of one steady step; the GPU runs beside it)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import triceratops_amd  # noqa: E402
from triceratops_amd import sharding, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
jobs = synth.toi_jobs(64, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
for s in range(3):
    torch.manual_seed(s)
    triceratops_amd.calc_probs_many(jobs)
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.manual_seed(7)
triceratops_amd.calc_probs_many(jobs)
torch.cuda.synchronize()
print("plain step %.4f s  %s" % (time.perf_counter() - t0, {k: round(v, 4) for k, v in sharding.timing.items()}))
pr = cProfile.Profile()
torch.manual_seed(8)
pr.enable()
triceratops_amd.calc_probs_many(jobs)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
