#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd SQLite or CSV) into a
small per-kernel table (markdown), for committing under profiles/.

usage: profiles/summarize.py <trace_results.db | kernel_stats.csv> > profiles/<name>.md
"""
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    return rows


def main():
    path = sys.argv[1]
    if path.endswith(".db"):
        rows = from_db(path)
    else:
        rows = []
        for r in csv.DictReader(open(path)):
            rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]), 0, 0, 0, 0))
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, calls, total, avg, mn, mx, vg, sg, lds, wg in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (short, calls, total / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot, vg, sg, lds, wg))


if __name__ == "__main__":
    main()
