run() { python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value'],1), {k:round(v,2) for k,v in s.items() if v>0.5})"; }
for a in "--batch 16" "--batch 20" "--batch 24" "--batch 28" "--batch 32" "--batch 40" "--batch 24 --contexts 4" "--batch 24 --contexts 2"; do
 echo "== $a"; run $a
done
