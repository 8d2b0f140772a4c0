#!/bin/bash
# Round-6 evidence in one call on the GPU box: the bench line, rocprofv3 kernel statistics of the same command (3 contexts) and of
# (the one-context runs take --opt shared_device=1 --opt scale_streams=0: the kernel forms of the headline, where the pool sets that option, one context at a time)
# one context alone (its real scales in order on one stream: option scale_streams = 0, so that a kernel's time is its own), and the two PMC traffic passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only).  Summaries -> gpurun_out/r06/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-repeats > $OUT/kt_bench.log 2>&1
python profiles/summarize.py $OUT/kt/trace_results.db > $OUT/kernel_stats.md
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kts -o trace -- python bench.py --contexts 1 --opt shared_device=1 --opt scale_streams=0 --batch 96 --steps 6 --warmup 2 --no-cpu-baseline --no-repeats > $OUT/solo_bench.log 2>&1
python profiles/summarize.py $OUT/kts/trace_results.db > $OUT/solo_kernel_stats.md
grep '^{"metric"' $OUT/solo_bench.log | tail -1 > $OUT/solo_bench.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python bench.py --contexts 1 --opt shared_device=1 --opt scale_streams=0 --batch 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-repeats > $OUT/pmc_$C.log 2>&1
done
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python profiles/make_traffic_json.py $F $W 96 > $OUT/pmc_traffic.json
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  timeout 400 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_sq -o pmc --output-format csv -- python bench.py --contexts 1 --opt shared_device=1 --opt scale_streams=0 --batch 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-repeats > $OUT/pmc_sq.log 2>&1
  FS=$(find $OUT/pmc_sq -name '*counter_collection.csv' | head -1)
  [ -n "$FS" ] && python profiles/pmc_dispatches.py "$FS" > $OUT/pmc_sq_largest_dispatch.csv
done
rm -rf $OUT/kt $OUT/kts $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
tail -c 1200 $OUT/bench.json; head -16 $OUT/kernel_stats.md | cut -c1-150; head -14 $OUT/solo_kernel_stats.md | cut -c1-150; cat $OUT/pmc_traffic.json | tail -30
# cfg 4 / cfg 5 lines, the depth cliff, the single-frame latency breakdown
python bench.py --config 4 --steps 6 --warmup 2 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python bench.py --config 5 --steps 6 --warmup 2 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python profiles/depth_cliff.py > $OUT/depth_cliff.json 2>/dev/null
python profiles/latency_breakdown.py > $OUT/latency.json 2>/dev/null
python profiles/ubench/level_seg_latency.py > $OUT/level_seg_latency.json 2>/dev/null
python profiles/ubench/corun_matrix.py > $OUT/corun_matrix.json 2>/dev/null
