"""Workload of profiles/draw_stats.sh: three passes over four synthetic TOIs (twelve lnZ_* calls each, N = 1e6,
device sampling, ONE stream so that nothing overlaps); run under rocprofv3 --kernel-trace, the script
prints nothing itself -- draw_stats.sh reads draw_kernel's and fill_kernel's durations from the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import triceratops_amd
from triceratops_amd import synth
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
triceratops_amd.set_sampling("device")
jobs = synth.toi_jobs(4, n_time=100, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
from triceratops_amd import sharding
sharding.streams = 1
for rep in range(3):
    np.random.seed(rep)
    triceratops_amd.calc_probs_many(jobs)
torch.cuda.synchronize()
