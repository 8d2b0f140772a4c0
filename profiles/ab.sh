#!/bin/bash
# generic A/B: every argument is an environment setting "A=1 B=2" (quote it); prints frames/s of the 3-context bench and the
# solo per-launch times of the kernels named in $KERNELS ($BENCH_FLAGS: more bench.py flags, e.g. --no-verify for timing
# experiments that change results)
out=${OUT:-gpurun_out/ab.txt}
: > $out
K=${KERNELS:-k_cascade_tile,k_level(fused),k_rank,k_smooth_vec}
for e in "$@"; do
  echo "== $e" >> $out
  env $e python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency $BENCH_FLAGS 2>/dev/null | K="$K" python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['solo']['kernels_ms_per_launch']
print(round(d['value']), {k:s[k] for k in os.environ['K'].split(',') if k in s})" >> $out 2>&1
done
cat $out
