cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_lv4; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4 > $OUT/bench.log 2>&1
python profiles/summarize.py $OUT/trace_results.db | cut -c1-140 | grep "k_level\|smooth\|tri_"
rm -rf $OUT
