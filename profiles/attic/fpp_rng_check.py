import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import triceratops_amd
from triceratops_amd import synth, fused
G = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
triceratops_amd.set_sampling("device")
for philox in (True, False):
    fused.PHILOX = philox
    res = []
    for seed in (1, 2, 3, 4):
        jobs = synth.toi_jobs(16, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(G, "trilegal_synth.csv"),
                              contrast_curve_file=os.path.join(G, "contrast_curve_synth.csv"))
        np.random.seed(seed); torch.manual_seed(seed)
        out = triceratops_amd.calc_probs_many(jobs)
        res.append([tg.FPP for tg in out])
    res = np.array(res)
    print("philox" if philox else "torch ", "mean FPP per seed", np.round(res.mean(axis=1), 4), " per-TOI std across seeds (median)", np.round(np.median(res.std(axis=0)), 4), " TOI 0..5 means", np.round(res.mean(axis=0)[:6], 4))
