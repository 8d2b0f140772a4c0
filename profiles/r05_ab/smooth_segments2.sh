# default against smooth_segments=1 (and the gradient launch alone), with the bench's three repeats; interleaved on one box
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(d.get('value_repeats'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec','k_level(fused)')})"; }
for i in 1 2; do
for a in "" "--opt smooth_segments=1" "--opt smooth_segments=2"; do
 echo "== $a"; run $a
done
echo "== GRAD_SEGMENTS=1"; ACF_HIP_GRAD_SEGMENTS=1 run
echo "== smooth_segments=1 GRAD_SEGMENTS=7"; ACF_HIP_GRAD_SEGMENTS=7 run --opt smooth_segments=1
done
