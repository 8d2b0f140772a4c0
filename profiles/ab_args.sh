#!/bin/bash
# A/B over bench.py argument sets (each argument one quoted string of flags): frames/s of each
out=${OUT:-gpurun_out/ab_args.txt}
: > $out
for a in "$@"; do
  echo "== $a" >> $out
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-verify $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'])" >> $out 2>&1
done
cat $out
