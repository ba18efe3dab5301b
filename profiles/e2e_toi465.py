"""End-to-end calc_probs() on TOI-465.01 (BASELINE configs[2]) for profiling:
python profiles/e2e_toi465.py [device|numpy-device|numpy] [real|blend] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pandas as pd, torch
import triceratops_amd
from triceratops_amd.triceratops import target
mode = sys.argv[1] if len(sys.argv) > 1 else "device"
tag = sys.argv[2] if len(sys.argv) > 2 else "blend"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "toi465_calc_probs.npz"))
cols = ("ID", "Tmag", "Jmag", "Hmag", "Kmag", "ra", "dec", "mass", "rad", "Teff", "plx", "fluxratio", "tdepth")
st = pd.DataFrame({c: g["%s_stars_%s" % (tag, c)] for c in cols})
st["ID"] = st["ID"].astype(np.int64)
triceratops_amd.set_sampling(mode)
kw = dict(contrast_curve_file=os.path.join(G, "toi465_cc.csv"), parallel=True, verbose=0)
for rep in range(reps + 1):
    tg = target(270380593, np.array([4]), stars=st.copy(), trilegal_fname=os.path.join(G, "trilegal_synth.csv"))
    np.random.seed(465); torch.manual_seed(465)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tg.calc_probs(g["time"], g["flux"], float(g["sigma"][0]), float(g["P_orb"][0]), N=20000 if rep == 0 else 1_000_000, **kw)
    torch.cuda.synchronize()
    if rep:
        print("%s sampling, %d scenarios, N=1e6: %.4f s  FPP=%.5f" % (mode, len(tg.lnZ), time.perf_counter() - t0, tg.FPP))
