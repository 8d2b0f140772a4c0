#!/bin/bash
cd $GRAFT_REPO_ROOT
ACF_HIP_SMOOTH_SIDE=1 ACF_HIP_FUSED_GRAD=2 timeout 900 python -m pytest tests/test_gpu_segments.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
: > gpurun_out/exp11.txt
for e in "A=1" "ACF_HIP_SMOOTH_SIDE=1" "A=2" "ACF_HIP_SMOOTH_SIDE=1"; do
  echo "== $e" >> gpurun_out/exp11.txt
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-profile --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('3ctx', round(d['value']))" >> gpurun_out/exp11.txt 2>&1
  env $e python bench.py --contexts 1 --batch 96 --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-profile --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1ctx', round(d['value']))" >> gpurun_out/exp11.txt 2>&1
done
cat gpurun_out/exp11.txt
