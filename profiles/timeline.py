#!/usr/bin/env python
"""Concurrency of a rocprofv3 kernel trace (csv): over the window spanned by the last N k_cascade_tile launches,
the fraction of time with 0, 1, 2, .. kernels running, and for every kernel name its summed duration and the mean
number of OTHER kernels running beside it."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"].replace("acfhip::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
tiles = [e for e in ev if "k_cascade_tile" in e[2]]
# the timed region: the last 12 tile launches (4 steps x 3 contexts)
t0 = tiles[-12][0] - 8_000_000
t1 = tiles[-1][1] + 1_000_000
ev = [e for e in ev if e[1] > t0 and e[0] < t1]
pts = []
for s, e, n in ev:
    pts.append((max(s, t0), 1))
    pts.append((min(e, t1), -1))
pts.sort()
hist = defaultdict(int)
cur, last = 0, t0
for t, d in pts:
    hist[cur] += t - last
    last = t
    cur += d
tot = t1 - t0
print("window %.2f ms; time with k kernels running:" % (tot / 1e6), {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per kernel name: total duration, mean concurrency beside it (integrate the count over its interval)
import bisect
times = [p[0] for p in pts]
cum = []
cur = 0
acc = 0
lastt = t0
integ = [0.0]
cnts = []
for t, d in pts:
    acc += cur * (t - lastt)
    integ.append(acc)
    lastt = t
    cur += d
    cnts.append(cur)
def integral(t):
    i = bisect.bisect_right(times, t)
    # integral up to times[i-1], then partial
    if i == 0:
        return 0.0
    base = integ[i]
    return base + cnts[i - 1] * (t - times[i - 1])
per = defaultdict(lambda: [0, 0.0, 0])
for s, e, n in ev:
    s, e = max(s, t0), min(e, t1)
    per[n][0] += e - s
    per[n][1] += integral(e) - integral(s)
    per[n][2] += 1
print("%-40s %8s %6s %6s" % ("kernel", "sum ms", "n", "conc"))
for n, (d, c, k) in sorted(per.items(), key=lambda x: -x[1][0])[:24]:
    print("%-40s %8.3f %6d %6.2f" % (n[:40], d / 1e6, k, c / max(d, 1)))
