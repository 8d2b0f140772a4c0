#!/bin/bash
# One call on the GPU box: kernel-trace stats + the two PMC traffic passes of the default bench command, summaries under gpurun_out/final/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/kt_bench.log 2>&1
python profiles/summarize.py $OUT/kt/trace_results.db > $OUT/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_$C.log 2>&1
done
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
B=$(python -c "import json;print(json.load(open('$OUT/bench.json'))['config']['frames_per_launch'])")
python profiles/make_traffic_json.py $F $W $B > $OUT/pmc_traffic.json
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
tail -c 1500 $OUT/bench.json; head -14 $OUT/kernel_stats.md | cut -c1-150; cat $OUT/pmc_traffic.json | head -30
