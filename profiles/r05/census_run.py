"""Trip census of cells_kernel on bench.py's workloads (library built with -DTRX_CENSUS: profiles/r05/isa_histogram.sh).
One step of 18 families x 1e5 rows; writes the per-launch means as JSON.
    TRX_LIB=profiles/ab_libs/libtrx_census.so python profiles/r05/census_run.py 2000 uniform > census_2000_uniform.json"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from triceratops_amd import _lib, synth  # noqa: E402

n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
grid = sys.argv[2] if len(sys.argv) > 2 else "uniform"
n_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
L = _lib.lib()
HAVE = hasattr(L, "trx_debug_census")          # (the product library has no census: the run then only launches the step,
if HAVE:                                       #  for rocprofv3 --pmc)
    L.trx_debug_census.restype = ctypes.c_int
    L.trx_debug_census.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
rng = np.random.default_rng(synth.SEED)
t = synth.time_grid(n_time)
if grid != "uniform":
    t = np.sort(rng.uniform(-0.25, 0.25, n_time))
t_d = _lib.dev(t)
curve, _ = _lib.flux_grid(_lib.MODEL_TP, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, synth.NSAMPLES, False)
f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))
rows = [_lib.dev(synth.family_rows(rng, fam, n_rows)) for fam in synth.FAMILIES]
out = torch.empty(n_rows, dtype=torch.float64, device="cuda")


def step():
    for (name, model, is_host, has_comp), r in zip(synth.FAMILIES, rows):
        flags = (_lib.FLAG_COMPANION_IS_HOST if is_host else 0) | _lib.FLAG_EVALUATE_EXCLUDED
        _lib.lnl_batch(model, flags, t_d, f_d, synth.SIGMA, r, synth.EXPTIME, synth.NSAMPLES, out=out)


step()
torch.cuda.synchronize()
if not HAVE:
    step()
    torch.cuda.synchronize()
    sys.exit(0)
buf = (ctypes.c_ulonglong * 32)()
_lib.check(L.trx_debug_census(buf, 1))
step()
_lib.check(L.trx_debug_census(buf, 1))
names = ["batch", "window_trip", "chunk0", "chunk1", "pass", "pair_trip", "pair_lanes", "flux_trip", "agm_trip", "kepler_full_pair",
         "kepler_full_plan", "flux_lanes", "contact_trip", "crossing_trip", "inside_trip"]
launches = len(synth.FAMILIES)
res = {n: buf[i] / launches for i, n in enumerate(names)}
res["cells_per_launch"] = float(n_rows) * n_time
res["n_time"], res["grid"], res["rows"] = n_time, grid, n_rows
print(json.dumps(res))
