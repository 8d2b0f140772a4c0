# k_smooth_grad_tri's acos table through L1 / L2 instead of 80 KB of LDS per workgroup (libacf_hip_acosg.so: -DACF_TRI_ACOS_GLOBAL): the thin chain gets slower, 96 CUs keep their LDS
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec',)})"; }
for i in 1 2 3; do echo "== table in LDS"; run; echo "== table through L1 / L2"; ACF_HIP_LIB=acf_amd/libacf_hip_acosg.so run; done
