# k_triy_chns's cells staged in LDS over four steps (64-byte stores; default build) against one 4-byte store per lane, channel and step
# (libacf_hip_direct.so: -DACF_TRIY_DIRECT): frames/s, the kernel alone, its WRITE_SIZE / FETCH_SIZE
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(d.get('value_repeats'), {k:round(v,3) for k,v in s.items() if k in ('k_triy_chns','k_level(fused)')})"; }
for i in 1 2 3; do
echo "== staged (default)"; run
echo "== direct"; ACF_HIP_LIB=acf_amd/libacf_hip_direct.so run
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
pmc staged A=1
pmc direct ACF_HIP_LIB=acf_amd/libacf_hip_direct.so
