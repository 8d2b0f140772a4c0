#!/bin/bash
# PMC pass (own run, kernel-trace only): usage profiles/run_pmc.sh <tag> "<counters>" [bench args]
TAG=$1; shift
CNT=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d $OUT -o pmc --output-format csv -- python bench.py "$@" --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1 || true
ls $OUT | head
F=$(find $OUT -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel," + ",".join(names))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get(names[0], 0))):
    print(k + "," + ",".join("%.4g" % (d[n] / max(cnt[(k, n)], 1)) for n in names))
PY

[ -n "$F" ] && python $GRAFT_REPO_ROOT/profiles/pmc_dispatches.py "$F" > $OUT/largest_dispatch.csv
find $OUT -name "*counter_collection.csv" -size +30M -delete
