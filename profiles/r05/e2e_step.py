"""A few 75-scenario calc_probs of TOI-465.01 at N = 1e6 for rocprofv3:   python profiles/r05/e2e_step.py [runs] [streams] [tag]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
from triceratops_amd import sharding  # noqa: E402
import test_toi465 as T  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2:
    sharding.streams = int(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else "blend"
triceratops_amd.set_sampling("device")
for s in range(runs + 1):
    torch.manual_seed(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T._run(tag, 1_000_000, s)
    torch.cuda.synchronize()
    print("run %d: %.2f ms" % (s, 1e3 * (time.perf_counter() - t0)), flush=True)
