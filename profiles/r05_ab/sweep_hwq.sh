run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-verify --no-repeats --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
for q in 2 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  for cfg in "--contexts 3 --batch 8" "--contexts 6 --batch 8" "--contexts 8 --batch 8" "--contexts 3 --batch 96" "--contexts 4 --batch 96"; do
    echo "== Q=$q $cfg"; run $cfg
  done
done
