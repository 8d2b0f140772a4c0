#!/bin/bash
# kernel-trace timeline of the 3-context bench: how many kernels run at any instant, idle time, and per-kernel
# (name) the time it spends alone vs beside 1, 2, .. other kernels
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT -o tl --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-verify $BENCH_ARGS > $OUT/bench.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python profiles/timeline.py $F
