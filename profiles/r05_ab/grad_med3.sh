# SV_GRAD's clamp of the acos index as one v_med3_f32 (4837 -> 4718 instructions per 16 columns of k_smooth_grad_tri) against the build before
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec',)})"; }
python -m pytest tests/test_gpu_segments.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do echo "== head"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run; echo "== med3"; run; done
