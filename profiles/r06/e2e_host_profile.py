"""Where the host's share of a 75-scenario calc_probs (TOI-465.01, N = 1e6) goes: cProfile over 40 runs, and the wall
time of a run split into calc_probs' own phases (sharding.last_timing)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
import test_toi465 as T  # noqa: E402

triceratops_amd.set_sampling("device")
for s in range(3):
    torch.manual_seed(s)
    T._run("blend", 1_000_000, s)
torch.cuda.synchronize()
ts = []
for s in range(20):
    torch.manual_seed(s)
    t0 = time.perf_counter()
    T._run("blend", 1_000_000, s)
    ts.append(1e3 * (time.perf_counter() - t0))
print("20 runs: best %.2f median %.2f ms" % (min(ts), sorted(ts)[10]))
pr = cProfile.Profile()
pr.enable()
for s in range(40):
    torch.manual_seed(s)
    T._run("blend", 1_000_000, s)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
