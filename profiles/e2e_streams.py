"""calc_probs() wall-clock on TOI-465.01 (N = 1e6, device sampling) against the number of HIP streams one
host thread deals the lnZ_* calls to, and against host threads: python profiles/e2e_streams.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, pandas as pd, torch
import triceratops_amd
from triceratops_amd import sharding
from triceratops_amd.triceratops import target
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "toi465_calc_probs.npz"))
cols = ("ID", "Tmag", "Jmag", "Hmag", "Kmag", "ra", "dec", "mass", "rad", "Teff", "plx", "fluxratio", "tdepth")
triceratops_amd.set_sampling("device")
kw = dict(contrast_curve_file=os.path.join(G, "toi465_cc.csv"), parallel=True, verbose=0)
for tag in ("blend", "real"):
    st = pd.DataFrame({c: g["%s_stars_%s" % (tag, c)] for c in cols}); st["ID"] = st["ID"].astype(np.int64)
    for thr, nst in ((1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (2, 1), (4, 1)):
        triceratops_amd.set_threads(thr)
        sharding.streams = nst
        best, enq = 9, 0
        for rep in range(5):
            tg = target(270380593, np.array([4]), stars=st.copy(), trilegal_fname=os.path.join(G, "trilegal_synth.csv"))
            np.random.seed(465); torch.manual_seed(465)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tg.calc_probs(g["time"], g["flux"], float(g["sigma"][0]), float(g["P_orb"][0]), N=20000 if rep == 0 else 1_000_000, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep and dt < best: best, enq = dt, sharding.timing["enqueue_s"]
        print("%s (%d scenarios) threads %d streams %d: %.4f s (host enqueue %.4f s)  FPP %.5f" % (tag, len(tg.lnZ), thr, nst, best, enq if thr == 1 else float("nan"), tg.FPP))
