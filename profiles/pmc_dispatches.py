#!/usr/bin/env python
"""Per-dispatch PMC rows of the largest dispatch of each kernel (by grid size) from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byd = collections.defaultdict(dict)
for r in rows:
    key = (r["Kernel_Name"][:48], r["Dispatch_Id"], int(r["Grid_Size"]))
    byd[key][r["Counter_Name"]] = byd[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
best = {}
for (k, d, g), v in byd.items():
    if k not in best or g > best[k][0]:
        best[k] = (g, v)
names = sorted({r["Counter_Name"] for r in rows})
w = csv.writer(sys.stdout)  # (kernel names hold commas: quoted)
w.writerow(["kernel", "grid"] + names)
for k, (g, v) in sorted(best.items(), key=lambda kv: -kv[1][1].get("SQ_WAVE_CYCLES", 0)):
    w.writerow([k, g] + ["%.4g" % v.get(n, 0) for n in names])
