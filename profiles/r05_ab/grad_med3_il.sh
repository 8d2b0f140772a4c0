# + the chain's first-column select gone (prev starts as the first input column) against the build before both changes
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec',)})"; }
python -m pytest tests/test_gpu_segments.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do echo "== head"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run; echo "== med3 + il"; run; done
