# k_triy_chns after the staged stores (158 VGPRs): buffer pitch 16 (49 KB per workgroup) and three waves per SIMD against the default (pitch 20, two waves)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(d.get('value_repeats'), {k:round(v,3) for k,v in s.items() if k in ('k_triy_chns',)})"; }
for i in 1 2; do
echo "== default (pitch 20, 2 waves)"; run
echo "== pitch 16, 2 waves"; ACF_HIP_LIB=acf_amd/libacf_hip_cbp16.so run
echo "== pitch 16, 3 waves"; ACF_HIP_LIB=acf_amd/libacf_hip_w3.so run
done
