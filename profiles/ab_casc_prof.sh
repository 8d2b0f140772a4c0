# per-kernel time of the tiled cascade for several stage-boundary settings (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in "$@"; do
 export ACF_HIP_CASC_BOUNDS=$B
 OUT=gpurun_out/prof_ab_$(echo $B | tr , _)
 rm -rf $OUT; mkdir -p $OUT
 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
 echo "== $B"; python profiles/summarize.py $OUT/trace_results.db | cut -c1-120 | grep -i "casc"
 rm -rf $OUT
done
