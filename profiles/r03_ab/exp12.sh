#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
export KERNELS="k_cascade_tile,k_level(fused),k_smooth_vec,k_tri_x,k_triy_chns,k_resample(image)"
OUT=gpurun_out/exp12_ab.txt bash profiles/ab.sh "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=1" "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=2"
