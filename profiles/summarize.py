#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd SQLite or CSV) into a
small per-kernel table (markdown), for committing under profiles/.

usage: profiles/summarize.py <trace_results.db | kernel_stats.csv> [--min-grid-frac F] > profiles/<name>.md

Launches are grouped by (kernel, grid size): the same kernel is launched once per real scale, and a bench run mixes
96-frame launches with the single-frame latency launches — one average over all of them describes none.
Groups of one kernel are listed largest grid first.
"""
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    if gx:
        g = "%s * %s * %s" % (gx, gx.replace("_x", "_y"), gx.replace("_x", "_z"))
    else:
        g = "0"
    rows = db.execute(
        "select name, %s as grid, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) from kernels group by name, grid" % g).fetchall()
    return rows


def main():
    path = sys.argv[1]
    if path.endswith(".db"):
        rows = from_db(path)
    else:
        rows = []
        for r in csv.DictReader(open(path)):
            rows.append((r["Name"], 0, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]), 0, 0, 0, 0))
    tot = sum(r[3] for r in rows) or 1
    per_name = {}
    for r in rows:
        per_name[r[0]] = per_name.get(r[0], 0) + r[3]
    rows.sort(key=lambda r: (-per_name[r[0]], r[0], -r[1]))
    print("| kernel | grid (threads) | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, grid, calls, total, avg, mn, mx, vg, sg, lds, wg in rows:
        short = name if len(name) < 80 else name[:77] + "..."
        print("| `%s` | %d | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (short, grid, calls, total / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot, vg, sg, lds, wg))


if __name__ == "__main__":
    main()
