run() { python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_batch']; print(round(d['value'],1), {k:round(v,2) for k,v in s.items() if v>0.3})"; }
for a in "--batch 192" "--batch 128" "--batch 256" "--batch 320" "--batch 384" "--batch 256 --contexts 4" "--batch 256 --contexts 2" "--batch 512 --contexts 2"; do
 echo "== $a"; run $a
done
