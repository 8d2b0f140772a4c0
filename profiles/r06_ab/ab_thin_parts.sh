#!/bin/bash
# A/B on one box: k_smooth_grad_tri as one workgroup per frame (thin_parts=1) against two row bands per frame (thin_parts=0: auto = 2 at 96 frames)
out=gpurun_out/ab_thin_parts.txt; : > $out
run() { echo "== $1" >> $out; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['value']), d['value_repeats'], 'verified', d['verified_frames'], 'tri solo ms', r['solo']['kernels_ms_per_batch'].get('k_smooth_grad_tri'), 'in-region', r['kernels_ms_per_step'].get('k_smooth_grad_tri'))" >> $out 2>&1; }
run "--opt thin_parts=1"
run "--opt thin_parts=0"
run "--opt thin_parts=1"
run "--opt thin_parts=0"
run "--opt thin_parts=0 --contexts 1"
run "--opt thin_parts=1 --contexts 1"
cat $out
