#!/bin/bash
cd $GRAFT_REPO_ROOT
: > gpurun_out/exp8.txt
for e in "ACF_HIP_NO_FUSED_GRAD=1" "A=1" "ACF_HIP_FUSED_GRAD_MINW=1000" "ACF_HIP_FUSED_GRAD_MINW=500" "ACF_HIP_NO_FUSED_GRAD=1" "ACF_HIP_FUSED_GRAD_MINW=1000"; do
  echo "== $e" >> gpurun_out/exp8.txt
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('3ctx', round(d['value']), 'lat1', round(d['latency_ms_batch1'],3), 'cfg3', round(d['config'].get('cfg3_64_frames_per_step_fps_1gpu',0)))" >> gpurun_out/exp8.txt 2>&1
  env $e python bench.py --contexts 1 --batch 96 --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-profile --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1ctx', round(d['value']))" >> gpurun_out/exp8.txt 2>&1
done
cat gpurun_out/exp8.txt
