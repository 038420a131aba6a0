"""From a rocprofv3 --kernel-trace CSV: how much kernel time overlaps (two kernels running at once), and which kernel pairs overlap.
usage: python tools/debug/overlap_from_trace.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[len(rows) // 2][0], rows[-1][1]     # second half of the run (timed steps)
rows = [r for r in rows if r[0] >= t0]
busy = sum(e - s for s, e, _, _ in rows)
span = t1 - t0
events = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
cur = 0; last = events[0][0]; hist = defaultdict(int)
for t, d in events:
    hist[cur] += t - last
    last = t; cur += d
print("span %.2f ms, sum of kernel durations %.2f ms; time with 0 / 1 / 2 / 3+ kernels running: %.2f / %.2f / %.2f / %.2f ms" % (
    span / 1e6, busy / 1e6, hist[0] / 1e6, hist[1] / 1e6, hist[2] / 1e6, sum(v for k, v in hist.items() if k >= 3) / 1e6))
queues = defaultdict(int)
for s, e, n, q in rows:
    queues[q] += e - s
print("kernel time per queue:", {k: round(v / 1e6, 2) for k, v in queues.items()})
pairs = defaultdict(int)
active = []
for s, e, n, q in rows:
    active = [a for a in active if a[1] > s]
    for a in active:
        pairs[(a[2], n)] += min(a[1], e) - s
    active.append((s, e, n, q))
for (a, b), v in sorted(pairs.items(), key=lambda t: -t[1])[:12]:
    print("  %8.3f ms  %s  ||  %s" % (v / 1e6, a, b))
