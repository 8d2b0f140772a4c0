#!/bin/bash
cd $GRAFT_REPO_ROOT
: > gpurun_out/exp15.txt
for e in "A=1" "ACF_HIP_RTILE_NW=4" "ACF_HIP_RTILE_NW=4 ACF_HIP_RTILE_TR=16" "ACF_HIP_RTILE_WG=2" "ACF_HIP_RTILE_WG=4" "A=2" "ACF_HIP_RTILE_NW=4 ACF_HIP_RTILE_WG=4"; do
  echo "== $e" >> gpurun_out/exp15.txt
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('3ctx', round(d['value']), 'tile solo', r['solo']['kernels_ms_per_launch'].get('k_cascade_tile'), 'in-region', {k:r['kernels_ms_per_step'][k] for k in ('k_cascade_tile','k_triy_chns','k_smooth_vec','k_tri_x')})" >> gpurun_out/exp15.txt 2>&1
done
cat gpurun_out/exp15.txt
