# gradMag + x pass on the smoothing chain at scale 1 as well (ACF_HIP_FUSED_GRAD_MINPX=500000: planes of >= 0.5 Mpx), scales 2 and 3 as kernels of their own
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d.get('verified_frames'))"; }
for i in 1 2 3; do echo "== scale 0 only"; run; echo "== scales 0 and 1"; ACF_HIP_FUSED_GRAD_MINPX=500000 run; done
