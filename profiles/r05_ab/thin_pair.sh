# the two thin smoothing launches of scale 0 side by side (colour planes on a side stream), eight hardware queues
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d.get('verified_frames'))"; }
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3; do
echo "== one stream"; run
echo "== pair"; ACF_HIP_THIN_PAIR=1 run
done
unset GPU_MAX_HW_QUEUES
echo "== default queues: one stream / pair"; run; ACF_HIP_THIN_PAIR=1 run
