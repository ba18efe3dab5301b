"""log-mean-exp bandwidth on a 3.2 GB vector: python profiles/lme_bw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from triceratops_amd import _lib
for name, fill in (("uniform(-3000,-1)", lambda x: x.uniform_(-3000.0, -1.0)), ("narrow(-40,-1)", lambda x: x.uniform_(-40.0, -1.0))):
    big = fill(torch.empty(400_000_000, dtype=torch.float64, device="cuda"))
    _lib.log_mean_exp(big, big.numel()); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): r = _lib.log_mean_exp(big, big.numel())
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("%-18s %.3f ms  %.0f GB/s  lnZ %.12f" % (name, ms, big.numel() * 8 / ms / 1e6, float(r.cpu()[0])))
    del big
