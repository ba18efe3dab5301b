"""GPU-busy time of the timed batch step of a rocprofv3 kernel trace (profiles/stats_batch.sh), attributed to the kernels
(an instant with k kernels running gives each 1/k) and split by the number of kernels running:
    python profiles/concurrency_levels.py <kernel_trace.csv> [draw launches per step = 768]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "draw_kernel" in r["Kernel_Name"]]
small = len(idx) - 3 * per
seg = rows[idx[small + per]:idx[small + 2 * per]]
ev = []
for k, r in enumerate(seg):
    ev.append((int(r["Start_Timestamp"]), 1, k))
    ev.append((int(r["End_Timestamp"]), 0, k))
ev.sort()
active, last, attr, conc = set(), ev[0][0], collections.Counter(), collections.Counter()
for t, kind, k in ev:
    if active:
        for a in active:
            attr[seg[a]["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]] += (t - last) / len(active)
        conc[len(active)] += t - last
    else:
        conc[0] += t - last
    last = t
    (active.add if kind else active.discard)(k)
tot = sum(attr.values())
print("timed step: span %.1f ms, busy %.1f ms; ms with k kernels running: %s" % (
    (ev[-1][0] - ev[0][0]) / 1e6, tot / 1e6, {k: round(v / 1e6, 1) for k, v in sorted(conc.items())}))
for n, v in attr.most_common(10):
    print("   %-46s %7.1f ms %5.1f %%" % (n, v / 1e6, 100 * v / tot))
