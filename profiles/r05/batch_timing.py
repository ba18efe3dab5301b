"""64-TOI step (BASELINE configs[3] on one GPU): wall-clock per step and where the host's enqueue time goes, with the launch
chains on and off, for a few stream counts.   python profiles/r05/batch_timing.py [tois] [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
from triceratops_amd import _lib, sharding, synth  # noqa: E402

tois = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
GOLD = os.path.join(ROOT, "tests", "golden")
jobs = synth.toi_jobs(tois, n_time=200, N=N, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
L = _lib.lib()


def step(seed):
    torch.manual_seed(seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    triceratops_amd.calc_probs_many(jobs)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for chain, streams, calls in ((1, 4, 12), (0, 4, 12), (1, 2, 12), (1, 3, 12), (1, 6, 12), (1, 8, 12), (1, 4, 6), (1, 4, 4), (1, 4, 12)):
    L.trx_set_star_chain(chain)
    sharding.streams = streams
    sharding.chain_calls = calls
    step(1)
    step(2)
    ts, tm = [], []
    for s in range(5):
        ts.append(step(10 + s))
        tm.append(dict(sharding.timing))
    k = int(np.argmin(ts))
    print("chain %d streams %d calls/piece %2d: step best %.4f mean %.4f s | enqueue %.4f (build %.4f library %.4f) wait %.4f prepare %.4f finish %.4f"
          % (chain, streams, calls, min(ts), np.mean(ts), tm[k]["enqueue_s"], tm[k]["build_s"], tm[k]["library_s"], tm[k]["wait_s"],
             tm[k]["prepare_s"], tm[k]["finish_s"]), flush=True)
