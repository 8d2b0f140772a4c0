# k_cascade_tile3: L2 prefetch of a later tile's footprint (ACF_HIP_TILE_PREFETCH = distance in tiles of the XCD's range) against the build without it
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_cascade_tile',)})"; }
for i in 1 2; do
echo "== head (no prefetch code)"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run
for pf in 0 16 32 64 96 160; do echo "== prefetch $pf"; ACF_HIP_TILE_PREFETCH=$pf run; done
done
echo "== persistent, one context: head / 0 / 64"
ACF_HIP_LIB=acf_amd/libacf_hip_head.so run --contexts 1 --persist 1
ACF_HIP_TILE_PREFETCH=0 run --contexts 1 --persist 1
ACF_HIP_TILE_PREFETCH=64 run --contexts 1 --persist 1
ACF_HIP_TILE_PREFETCH=3 run --contexts 1 --persist 1
