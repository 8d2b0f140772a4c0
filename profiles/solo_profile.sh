#!/bin/bash
# kernel statistics of ONE detector context at the bench's per-launch batch (96 frames): each kernel alone on the machine
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/solo; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o trace -- python bench.py --contexts 1 --batch 96 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
python profiles/summarize.py $OUT/kt/trace_results.db > $OUT/kernel_stats.md
tail -1 $OUT/bench.log > $OUT/bench.json
rm -rf $OUT/kt
head -16 $OUT/kernel_stats.md | cut -c1-160
