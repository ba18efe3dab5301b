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
# the configurations are visited in turn, several rounds (a box's clocks drift over a run: interleaved, every
# configuration sees the same conditions); best and median step per configuration
import statistics
configs = (1, 2, 3, 4, 6, 8)
runs = {c: [] for c in configs}
for rep in range(7):
    for streams in configs:
        sharding.streams = streams
        np.random.seed(5 + rep)
        torch.manual_seed(5 + rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        triceratops_amd.calc_probs_many(jobs)
        torch.cuda.synchronize()
        runs[streams].append((time.perf_counter() - t0, dict(sharding.timing)))
for streams in configs:
    r = sorted(runs[streams], key=lambda q: q[0])
    best, tm = r[0]
    med = statistics.median(q[0] for q in r)
    print("streams %d: step best %.3f s median %.3f s  (best: enqueue %.3f s  wait %.3f s  prepare %.4f s  finish %.4f s  other %.3f s)" % (
        streams, best, med, tm["enqueue_s"], tm["wait_s"], tm["prepare_s"], tm["finish_s"],
        best - tm["enqueue_s"] - tm["wait_s"] - tm["prepare_s"] - tm["finish_s"]))
