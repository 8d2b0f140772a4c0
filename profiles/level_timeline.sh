#!/bin/bash
# kernel-trace timeline of one step: which k_level launches overlap, and how long each lasts
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lvtl
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT -o tl --output-format csv -- python bench.py --batch ${1:-64} --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
python - $OUT/tl_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: take the last k_cascade_tile and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "k_cascade_tile" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].replace("acfhip::", "").split("(")[0][:34]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-36s q%-3s start %8.3f ms  dur %7.3f ms  grid %s" % (n, r.get("Queue_Id", "?"), s / 1e6, (e - s) / 1e6, r.get("Grid_Size", "")))
PY
