#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        per = defaultdict(float)
        for row in csv.DictReader(fh):
            key = (row["Dispatch_Id"], row["Kernel_Name"], row["Counter_Name"])
            per[key] += float(row["Counter_Value"])
        for (d, k, c), v in per.items():
            acc[k][c].append(v)
for k in sorted(acc, key=lambda k: -len(acc[k])):
    short = k.split("(")[0][-60:]
    if not any(s in k for s in ("rowc_kernel", "cells_kernel", "draw_kernel", "lme", "chi2")):
        continue
    print("==", ("cells_kernel" if "cells_kernel" in k else "rowc_kernel" if "rowc_kernel" in k else short), "<0" if "<0" in k else "<1" if "<1" in k else "")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
