run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value']), {k:round(v,3) for k,v in s.items() if k in ('k_smooth_vec','k_level(fused)')})"; }
for a in "" "--opt smooth_segments=1" "--opt smooth_segments=2" "--opt smooth_segments=3" "--opt smooth_segments=4" "--opt smooth_segments=6" "--opt smooth_segments=8" "--opt smooth_segments=3 --opt smooth_warm=32" "--opt smooth_warm=32" "--opt smooth_warm=64"; do
 echo "== $a"; run $a
done
