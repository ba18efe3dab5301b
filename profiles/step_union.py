"""GPU busy time (union of the kernel intervals) and per-kernel sums of each batch step in a rocprofv3 kernel
trace of `bench.py --mode batch --steps 1 --warmup 1` (profiles/stats_batch.sh):
    python profiles/step_union.py <kernel_trace.csv> [draw launches per step = 768]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "draw_kernel" in r["Kernel_Name"]]
small = len(idx) - 3 * per
bounds = [idx[small], idx[small + per], idx[small + 2 * per], len(rows)]
for s, what in enumerate(("warm-up step", "timed step", "traced step (torch chain, one stream)")):
    seg = rows[bounds[s]:bounds[s + 1]]
    t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
    tot, cnt = collections.Counter(), collections.Counter()
    for r in seg:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]
        tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[n] += 1
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for a, b in iv[1:]:
        if a > ce:
            busy += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    print("%s: span %.1f ms, kernel sum %.1f ms, GPU busy (union) %.1f ms, %d launches" % (
        what, (t1 - t0) / 1e6, sum(tot.values()), busy / 1e6, len(seg)))
    for n, v in tot.most_common(8):
        print("   %-52s %5d %8.1f ms" % (n, cnt[n], v))
