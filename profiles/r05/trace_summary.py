"""Per-step summary of a rocprofv3 kernel trace of profiles/r05/batch_step.py / e2e_step.py: launches, kernel-time sum,
GPU busy time (union of the kernel intervals), kernels in flight, time per kernel.
    python profiles/r05/trace_summary.py <kernel_trace.csv> <marker kernel fragment, one per chain>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")) for r in rows]
# steps: separated by gaps of more than 3 ms without any kernel
steps, cur, last_end = [], [], None
for s, e, n in ev:
    if last_end is not None and s - last_end > 3_000_000 and cur:
        steps.append(cur)
        cur = []
    cur.append((s, e, n))
    last_end = e if last_end is None else max(last_end, e)
if cur:
    steps.append(cur)
print("%d dispatches, %d bursts separated by > 3 ms idle" % (len(ev), len(steps)))
for k, seg in enumerate(steps):
    if len(seg) < 50:
        continue
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    iv = sorted((s, e) for s, e, _ in seg)
    busy, (cs, ce) = 0, iv[0]
    for a, b in iv[1:]:
        if a > ce:
            busy += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    pts = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
    lvl, lastt, hist = 0, None, collections.Counter()
    for t, d in pts:
        if lastt is not None:
            hist[lvl] += t - lastt
        lvl += d
        lastt = t
    tot, cnt = collections.Counter(), collections.Counter()
    for s, e, n in seg:
        key = n.split("(")[0][-34:]
        tot[key] += (e - s) / 1e6
        cnt[key] += 1
    span = (t1 - t0) / 1e6
    print("burst %d: span %.1f ms, %d launches (%d of them copies / fills), kernel-time sum %.1f ms, GPU busy %.1f ms; in flight: %s"
          % (k, span, len(seg), sum(v for n, v in cnt.items() if "rocclr" in n or "FillFunctor" in n), sum(tot.values()), busy / 1e6,
             ", ".join("%d: %.0f %%" % (l, 100.0 * v / max(sum(hist.values()), 1)) for l, v in sorted(hist.items()) if v > 0.005 * sum(hist.values()))))
    for n, v in tot.most_common(10):
        print("      %-36s %5d launches %8.2f ms  (%.1f us each)" % (n, cnt[n], v, 1e3 * v / cnt[n]))
