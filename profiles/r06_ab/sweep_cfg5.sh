#!/bin/bash
# cfg 5 (4K, LDCF k = 4): contexts x frames per launch, turns
out=gpurun_out/sweep_cfg5.txt; : > $out
run() { echo "== $1" >> $out; python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-verify $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['value_repeats'])" >> $out 2>&1; }
run "--contexts 3 --batch 24"
run "--contexts 2 --batch 36"
run "--contexts 4 --batch 18"
run "--contexts 3 --batch 32"
run "--contexts 3 --batch 24 --turns 0"
run "--contexts 3 --batch 24 --turns 1"
run "--contexts 3 --batch 24 --opt shared_device=0"
run "--contexts 3 --batch 24 --opt scale_streams=1"
run "--contexts 3 --batch 24 --opt fused_grad=0"
run "--contexts 3 --batch 24 --opt level_segments=1"
run "--contexts 3 --batch 24"
cat $out
