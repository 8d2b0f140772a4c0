#!/bin/bash
# the two smoothing launches of a scale with their own segment counts
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_segments.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
export KERNELS="k_smooth_vec"
OUT=gpurun_out/exp17_ab.txt bash profiles/ab.sh "ACF_HIP_SMOOTH_SEGMENTS=4" "A=1" "ACF_HIP_GRAD_SEGMENTS=6" "ACF_HIP_GRAD_SEGMENTS=8" "ACF_HIP_GRAD_SEGMENTS=4" "ACF_HIP_SMOOTH_SEGMENTS=4" "A=2"
