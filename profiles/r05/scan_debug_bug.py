import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import triceratops_amd
from triceratops_amd import _lib, fused, sharding
import anchors
L = _lib.lib()
for case in ("toi411", "toi465_nocc"):
    for N in range(150_000, 420_000, 10_000):
        out = []
        for bug in (0, 1):
            L.trx_set_debug_bug(bug)
            try:
                lnZ, prob, fpp, rp = anchors.run(case, 7, N=N, sampling="device")
                out.append("ok FPP %.6f" % fpp)
            except _lib.TrxError as e:
                out.append("TrxError")
        print(case, N, out, flush=True)
L.trx_set_debug_bug(0)
