"""bounded evaluation on / off / torch-operator path on TOI-465.01: per-scenario lnZ must agree bit for bit"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import anchors, triceratops_amd
from triceratops_amd import fused, sharding, _lib
from triceratops_amd.triceratops import target
triceratops_amd.set_sampling("device")
sharding.per_unit_seed = True
L = _lib.lib()
case = sys.argv[1] if len(sys.argv) > 1 else "toi465_cc"
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
c = anchors.CASES[case]; stars, t, f, sigma, P = anchors.inputs(case)
res = {}
for name, native, prune in (("pruned", True, 1), ("full", True, 0), ("torch", False, 0), ("pruned2", True, 1)):
    fused.NATIVE = native; L.trx_set_bounded_evaluation(prune)
    tg = target(c["ID"], np.array([1]), mission=c["mission"], stars=stars, trilegal_fname=anchors.TRILEGAL)
    np.random.seed(7); torch.manual_seed(7)
    n = ctypes_n = None
    import ctypes
    cnt = ctypes.c_ulonglong(0); L.trx_pruned_rows(ctypes.byref(cnt), 1)
    tg.calc_probs(t, f, sigma, P, contrast_curve_file=c["cc"], N=N, parallel=True, verbose=0)
    L.trx_pruned_rows(ctypes.byref(cnt), 1)
    res[name] = np.array(tg.lnZ)
    print("%-8s FPP %.6g  pruned rows %d  lnZ[:4] %s" % (name, tg.FPP, cnt.value, np.array2string(res[name][:4], precision=6)))
for a, b in (("pruned", "full"), ("full", "torch"), ("pruned", "pruned2")):
    same = (res[a] == res[b]) | (np.isnan(res[a]) & np.isnan(res[b]))
    print(a, "==", b, ":", bool(same.all()))
    for k in np.argwhere(~same).ravel():
        print("   scenario %d (%s): %s %.17g  %s %.17g" % (k, anchors.SCENARIOS[k] if k < 15 else "N", a, res[a][k], b, res[b][k]))
