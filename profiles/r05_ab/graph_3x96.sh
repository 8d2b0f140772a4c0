# the 96-frame call replayed as a captured HIP graph (option graph = 1) in the 3-context headline
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-repeats "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d.get('verified_frames'))"; }
for i in 1 2 3; do echo "== launches"; run; echo "== graph"; run --opt graph=1; done
