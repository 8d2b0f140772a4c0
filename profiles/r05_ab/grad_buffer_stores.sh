# k_smooth_grad_tri: M and U through buffer stores (descriptor + fixed lane offset + scalar column offset) against the build before this and the
# two small changes before it (libacf_hip_head.so: 4837 instructions per 16 columns).  (O through a second descriptor gave wrong orientation
# bins for rows >= ~970 — waves 4 and up — and no explanation in the ISA: O keeps its pointer store.)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec',)})"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for s in 401 402; do python tests/fuzz_parity.py $s 120 2>&1 | tail -1; done
for i in 1 2 3; do echo "== head"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run; echo "== buffer stores (M, U)"; run; done
