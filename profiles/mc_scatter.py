"""Monte-Carlo scatter of lnZ: host (numpy stream) vs device (Philox) sampling, several seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import triceratops_amd
from triceratops_amd import marginal_likelihoods as ml
g = np.load(os.path.join(ROOT, "tests", "golden", "lnz_cases.npz"))
t, f, sigma = g["time"], g["flux"], float(g["sigma"][0])
base = (t, f, sigma, 3.3, 0.82, 0.8, 5100.0)
for N in (400_000, 4_000_000):
    for mode in ("numpy", "device"):
        triceratops_amd.set_sampling(mode)
        vals = []
        for seed in range(6 if (mode == "device" or N < 1e6) else 3):
            np.random.seed(seed); torch.manual_seed(seed)
            vals.append(ml.lnZ_TTP(*base, 0.0, N, True)["lnZ"])
        print("TTP N=%8d %-6s" % (N, mode), " ".join("%.3f" % v for v in vals), " mean %.3f std %.3f" % (np.mean(vals), np.std(vals)))
triceratops_amd.set_sampling("numpy")
