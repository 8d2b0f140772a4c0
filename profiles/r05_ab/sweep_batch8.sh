for cfg in "--contexts 1 --batch 8" "--contexts 2 --batch 8" "--contexts 3 --batch 8" "--contexts 4 --batch 8" "--contexts 6 --batch 8" "--contexts 8 --batch 8" "--contexts 3 --batch 16" "--contexts 3 --batch 32"; do
  echo "== $cfg"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-verify --no-repeats --no-profile $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"
done
