import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import triceratops_amd
from triceratops_amd import _lib, sharding, synth
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
GOLD = os.path.join(ROOT, "tests", "golden")
triceratops_amd.set_sampling("device")
ntoi = int(sys.argv[1]); streams = int(sys.argv[2]); N = int(float(sys.argv[3]))
sharding.streams = streams
jobs = synth.toi_jobs(ntoi, n_time=200, N=N, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
bad = 0
for s in range(3):
    torch.manual_seed(s)
    for k in range(ntoi):
        try:
            triceratops_amd.calc_probs_many(jobs[k:k + 1])
        except Exception as e:
            bad += 1
            print("seed", s, "toi", k, "FAILED:", str(e)[:90])
print("ntoi", ntoi, "streams", streams, "N", N, "failures", bad)
