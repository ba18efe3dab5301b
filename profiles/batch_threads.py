"""64-TOI step against the number of host threads (one stream each) and against streams of one thread:
    python profiles/batch_threads.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import triceratops_amd
from triceratops_amd import sharding, synth
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
triceratops_amd.set_sampling("device")
tri, cc = os.path.join(GOLD, "trilegal_synth.csv"), os.path.join(GOLD, "contrast_curve_synth.csv")
jobs = synth.toi_jobs(64, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
small = synth.toi_jobs(2, n_time=200, N=20000, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
triceratops_amd.calc_probs_many(small)
configs = [("threads", 1, 3), ("threads", 1, 4), ("threads", 1, 5), ("threads", 2, 0), ("threads", 3, 0), ("threads", 4, 0), ("threads", 6, 0)]
runs = {c: [] for c in configs}
for rep in range(6):
    for c in configs:
        sharding.threads, sharding.streams = c[1], (c[2] or sharding.streams)
        np.random.seed(5 + rep); torch.manual_seed(5 + rep); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = triceratops_amd.calc_probs_many(jobs)
        torch.cuda.synchronize()
        runs[c].append(time.perf_counter() - t0)
for c in configs:
    print("threads %d streams %s: step best %.3f s median %.3f s" % (c[1], c[2] or "one per thread", min(runs[c]), statistics.median(runs[c])))
sharding.threads = 1
