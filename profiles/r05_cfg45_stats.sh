#!/bin/bash
# rocprofv3 kernel statistics of cfg 4 and cfg 5 with one context alone (a kernel's time is its own) -> gpurun_out/r05c/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
cd $R
for C in 4 5; do
  ACF_HIP_SCALES_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt$C -o trace -- python bench.py --config $C --contexts 1 --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-repeats > $OUT/cfg${C}_solo.log 2>&1
  python profiles/summarize.py $OUT/kt$C/trace_results.db > $OUT/cfg${C}_solo_kernel_stats.md
  rm -rf $OUT/kt$C
done
head -20 $OUT/cfg4_solo_kernel_stats.md | cut -c1-140
