#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_segments.py tests/test_gpu_configs.py tests/test_golden.py -m gpu -x -q > gpurun_out/exp9_pytest.log 2>&1
tail -5 gpurun_out/exp9_pytest.log
: > gpurun_out/exp9.txt
for e in "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=1" "ACF_HIP_FUSED_GRAD=2" "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=2" "ACF_HIP_FUSED_GRAD_MINPX=400000"; do
  echo "== $e" >> gpurun_out/exp9.txt
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['solo']['kernels_ms_per_launch']
print('3ctx', round(d['value']), 'lat1', round(d['latency_ms_batch1'],3), 'cfg3', round(d['config'].get('cfg3_64_frames_per_step_fps_1gpu',0)), {k:s[k] for k in ('k_smooth_vec','k_grad_mag') if k in s})" >> gpurun_out/exp9.txt 2>&1
done
cat gpurun_out/exp9.txt
