#!/usr/bin/env python3
"""tests/golden/reference_runs.npz: results of the REFERENCE's own calc_probs at N = 1e6 on the notebook
inputs, run on the CPU of the build container by profiles/reference_fpp_cpu.py (the reference imported from
/root/reference under the shims of make_golden.py: oracle QuadraticModel at the pytransit seam, synthetic
TRILEGAL table).  ~80-100 s per run, hence few runs (16 of TOI-465.01, 16 of TOI-411.02); they are the sample tests/test_gpu_notebook_anchors.py
compares the device path's distribution with (same code base as the device path mirrors, unlike the stored
notebook outputs, which an older release produced).

    python profiles/reference_fpp_cpu.py toi465_nocc 6 >  profiles/r04/reference_fpp_cpu.txt
    python profiles/reference_fpp_cpu.py toi411 4      >> profiles/r04/reference_fpp_cpu.txt      (round 3: seeds 1000-1003)
    python profiles/reference_fpp_cpu.py toi411 12 1000000 1004 >> profiles/r04/reference_fpp_cpu.txt   (round 4: 1004-1015)
    python profiles/reference_fpp_cpu.py toi465_nocc 10 1000000 1006 >> profiles/r04/reference_fpp_cpu.txt   (round 4: 1006-1015)
    python tests/golden/make_reference_runs.py profiles/r04/reference_fpp_cpu.txt
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
pat = re.compile(r"^(\w+) seed (\d+) N (\d+): FPP ([\d.eE+-]+)\s+TP ([\d.eE+-]+) PTP ([\d.eE+-]+) STP ([\d.eE+-]+) "
                 r"DTP ([\d.eE+-]+)\s+lnZ TP ([\d.eE+-]+) PTP ([\d.eE+-]+) STP ([\d.eE+-]+)\s+Rp ([\d.eE+-]+)")
runs = {}
for fname in sys.argv[1:]:
    for line in open(fname):
        m = pat.match(line)
        if m:
            runs.setdefault(m.group(1), []).append([float(x) for x in m.groups()[1:]])
out = {}
for case, rows in runs.items():
    a = np.array(rows)
    out[case + "_seed"], out[case + "_N"] = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    out[case + "_FPP"] = a[:, 2]
    out[case + "_prob"] = a[:, 3:7]          # TP PTP STP DTP
    out[case + "_lnZ"] = a[:, 7:10]          # TP PTP STP
    out[case + "_Rp"] = a[:, 10]
    print(case, len(rows), "runs: FPP", a[:, 2])
np.savez_compressed(os.path.join(HERE, "reference_runs.npz"), **out)
