for cfg in "--contexts 3 --batch 96" "--contexts 4 --batch 72" "--contexts 4 --batch 96" "--contexts 3 --batch 128" "--contexts 2 --batch 144" "--contexts 3 --batch 96 --turns 0" "--contexts 3 --batch 96 --turns 1" "--contexts 3 --batch 96 --persist 1" "--contexts 6 --batch 48"; do
  echo "== $cfg"; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))"
done
