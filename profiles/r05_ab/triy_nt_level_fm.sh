# (1) k_triy_chns's M / O / U reads non-temporal (default build) against plain loads (libacf_hip_plainld.so: -DACF_TRIY_PLAIN_LOADS): frames/s and
#     the kernel's WRITE_SIZE / FETCH_SIZE; (2) k_level_all dispatched frame-group-major (ACF_HIP_LEVEL_FRAME_MAJOR=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(d.get('value_repeats'), {k:round(v,3) for k,v in s.items() if k in ('k_triy_chns','k_level(fused)')})"; }
for i in 1 2; do
echo "== nt loads (default)"; run
echo "== plain loads"; ACF_HIP_LIB=acf_amd/libacf_hip_plainld.so run
echo "== level frame-major"; ACF_HIP_LEVEL_FRAME_MAJOR=1 run
done
pmc() { # $1 = tag, rest = env
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$1_$C
    env "${@:2}" ACF_HIP_SCALES_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$1_$C -o pmc --output-format csv -- python bench.py --contexts 1 --opt shared_device=1 --batch 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-repeats > /tmp/pmc_$1_$C.log 2>&1
  done
  F=$(find /tmp/pmc_$1_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find /tmp/pmc_$1_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  python profiles/make_traffic_json.py $F $W 96 | python -c "
import json,sys; t=json.load(sys.stdin); print('$1', {k:(t['fetch_MB_per_frame'][k], t['write_MB_per_frame'][k]) for k in ('k_triy_chns','k_level(fused)','k_smooth_vec','k_cascade_tile')}, t['total_MB_per_frame'])"
}
pmc nt A=1
pmc plain ACF_HIP_LIB=acf_amd/libacf_hip_plainld.so
pmc levelfm ACF_HIP_LEVEL_FRAME_MAJOR=1
