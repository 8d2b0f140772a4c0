# stage E's (packed, 12-byte) node records requested ahead of the sparse stage against the build before (libacf_hip_head.so)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_cascade_tile',)})"; }
for i in 1 2 3; do
echo "== head"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run
echo "== early packed E records"; run
done
python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -m gpu -x -q -k "cascade or cfg" 2>&1 | tail -3
