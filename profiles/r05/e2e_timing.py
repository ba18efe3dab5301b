"""TOI-465.01 calc_probs (BASELINE configs[2]: 75 scenarios, N = 1e6; and the 15-scenario table): wall-clock with the launch
chains on / off, stream counts and calls per piece.   python profiles/r05/e2e_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
from triceratops_amd import _lib, sharding  # noqa: E402
import test_toi465 as T  # noqa: E402

triceratops_amd.set_sampling("device")
L = _lib.lib()


def run(tag, seed):
    torch.manual_seed(seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tg = T._run(tag, 1_000_000, seed)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, tg


for tag in ("blend", "real"):
    for chain, streams, calls in ((1, 4, 12), (0, 4, 12), (1, 1, 16), (1, 2, 12), (1, 3, 12), (1, 6, 12), (1, 8, 8), (1, 4, 8), (1, 4, 6), (1, 4, 16), (1, 6, 8)):
        L.trx_set_star_chain(chain)
        sharding.streams = streams
        sharding.chain_calls = calls
        run(tag, 1)
        run(tag, 2)
        ts, tm = [], []
        for s in range(7):
            dt, tg = run(tag, 10 + s)
            ts.append(dt)
            tm.append(dict(sharding.timing))
        k = int(np.argmin(ts))
        print("%s chain %d streams %d calls/piece %2d: best %.2f median %.2f ms | enqueue %.2f (build %.2f library %.2f) wait %.2f ms"
              % (tag, chain, streams, calls, 1e3 * min(ts), 1e3 * np.median(ts), 1e3 * tm[k]["enqueue_s"], 1e3 * tm[k]["build_s"],
                 1e3 * tm[k]["library_s"], 1e3 * tm[k]["wait_s"]), flush=True)
