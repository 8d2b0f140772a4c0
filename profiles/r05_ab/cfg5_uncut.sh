# cfg 5 (3 x 24 frames of 4K: below shared_device's 64 frames): uncut smoothing chains forced
run() { python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
for i in 1 2; do echo "== default"; run; echo "== smooth_segments=1"; run --opt smooth_segments=1; done
