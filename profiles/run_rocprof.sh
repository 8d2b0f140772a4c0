#!/bin/bash
# Run on the GPU box via gpurun: kernel-trace stats of the bench command.
# usage: profiles/run_rocprof.sh <tag> [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py "$@" --no-cpu-baseline > $OUT/bench.log 2>&1 || true
tail -2 $OUT/bench.log
find $OUT -name '*stats*' | head
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp $F $OUT/kernel_stats.csv && head -30 $F
# keep only small summaries for merging back
find $OUT -name '*kernel_trace.csv' -size +20M -delete || true
