# convTri's x pass on the gradient plane's smoothing chain (k_smooth_grad_tri) against k_tri_x5v, with and without the colour planes' segments
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(d.get('value_repeats'), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec','k_tri_x','k_triy_chns')})"; }
for i in 1 2; do
for a in "--opt fused_tri=0" "--opt fused_tri=1" "--opt fused_tri=0 --opt smooth_segments=1" "--opt fused_tri=1 --opt smooth_segments=1"; do
 echo "== $a"; run $a
done
done
