cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for D in 0 1 2 3; do
 export ACF_HIP_CASC_DEBUG=$D
 OUT=gpurun_out/prof_dbg_$D
 rm -rf $OUT; mkdir -p $OUT
 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
 echo "== debug $D"; python profiles/summarize.py $OUT/trace_results.db | cut -c1-120 | grep -i "casc"
 rm -rf $OUT
done
