# two tile workgroups + one level workgroup per CU by construction (tile LDS padded by 3 KB: three do not fit, two leave 52.5 KB) with the turns off / on
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
echo "== default (turns 5)"; run
for t in 0 1 5; do echo "== PAD 3 KB, turns $t"; ACF_HIP_TILE_PAD_KB=3 run --turns $t; done
for t in 0 1; do echo "== no pad, turns $t"; run --turns $t; done
echo "== PAD 4 KB, turns 0"; ACF_HIP_TILE_PAD_KB=4 run --turns 0
echo "== default (turns 5)"; run
