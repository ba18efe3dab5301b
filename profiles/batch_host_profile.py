"""cProfile of the host side of one batch-mode step (64 TOIs, device sampling, one thread, 3 streams):
    python profiles/batch_host_profile.py [tois] [N]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import triceratops_amd
from triceratops_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
tois = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
triceratops_amd.set_sampling("device")
tri, cc = os.path.join(GOLD, "trilegal_synth.csv"), os.path.join(GOLD, "contrast_curve_synth.csv")
jobs = synth.toi_jobs(tois, n_time=200, N=N, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
small = synth.toi_jobs(2, n_time=200, N=20000, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
triceratops_amd.calc_probs_many(small)
np.random.seed(3)
triceratops_amd.calc_probs_many(jobs)
torch.cuda.synchronize()
pr = cProfile.Profile()
np.random.seed(4)
pr.enable()
triceratops_amd.calc_probs_many(jobs)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(35)
