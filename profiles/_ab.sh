out=gpurun_out/ab_tile.txt; : > $out
K="k_smooth_vec,k_cascade_tile"
run() { echo "== $1 $2" >> $out; env $1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency $2 2>/dev/null | K="$K" python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['solo']['kernels_ms_per_launch']
print(round(d['value']), {k:s[k] for k in os.environ['K'].split(',') if k in s}, d.get('verified_frames'))" >> $out 2>&1; }
for i in 1 2; do
run "ACF_HIP_LIB=acf_amd/libacf_hip_head.so" ""
run "X=1" ""
done
cat $out
