run() { python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value'],1), {k:round(v,2) for k,v in s.items() if v>0.5})"; }
for a in "" "--opt level_segments=2" "--opt level_segments=3" "--opt level_segments=4" "--opt level_segments=6" "--batch 8 --contexts 6" "--batch 24" "--batch 32 --contexts 2" "--contexts 4 --batch 12" "--opt scale_streams=1"; do
 echo "== $a"; run $a
done
