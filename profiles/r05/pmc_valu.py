"""mean of a few SQ counters per launch of the kernels whose name contains a fragment, from rocprofv3 --pmc csv output
    python profiles/r05/pmc_valu.py <dir> <fragment>"""
import collections
import csv
import glob
import sys

d, frag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(tot.items()):
    print("%-28s launches %3d  mean %.4e" % (k, len(v), sum(v) / len(v)))
