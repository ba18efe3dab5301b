"""draw_kernel time per scenario: one calc_probs-like pass of the ten lnZ_* calls at N = 1e6 (device sampling),
kernel durations from HIP events around each call would include the likelihood, so this script runs the draw
kernel alone through trx_draw_scenario's counted variant -- via fused.DUMP-free calls under rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 profiles/draw_times.py
and prints nothing itself; see profiles/draw_stats.sh."""
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
