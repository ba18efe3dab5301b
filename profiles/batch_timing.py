"""Host enqueue / stream wait split of a batch-mode step (sharding.timing) for a few stream counts:
    python profiles/batch_timing.py [tois] [N]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import triceratops_amd
from triceratops_amd import sharding, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
tois = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
triceratops_amd.set_sampling("device")
tri, cc = os.path.join(GOLD, "trilegal_synth.csv"), os.path.join(GOLD, "contrast_curve_synth.csv")
jobs = synth.toi_jobs(tois, n_time=200, N=N, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
small = synth.toi_jobs(2, n_time=200, N=20000, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
triceratops_amd.calc_probs_many(small)
for streams in (1, 2, 3, 4, 6, 8, 12):
    sharding.streams = streams
    best = None
    for rep in range(5):
        np.random.seed(5 + rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        triceratops_amd.calc_probs_many(jobs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, dict(sharding.timing))
    print("streams %d: step %.3f s  enqueue %.3f s  wait %.3f s  other %.3f s" % (
        streams, best[0], best[1]["enqueue_s"], best[1]["wait_s"], best[0] - best[1]["enqueue_s"] - best[1]["wait_s"]))
