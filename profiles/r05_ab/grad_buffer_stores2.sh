# k_smooth_grad_tri: M, O, U through buffer stores, O through M's descriptor + the planes' distance as scalar offset; against the committed build (libacf_hip_head.so)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), d.get('verified_frames'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec',)})"; }
python profiles/r05_ab/dbg/dbg_tri.py 2>&1 | tail -4
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for s in 501 502; do python tests/fuzz_parity.py $s 120 2>&1 | tail -1; done
for i in 1 2 3; do echo "== committed"; ACF_HIP_LIB=acf_amd/libacf_hip_head.so run; echo "== buffer stores"; run; done
