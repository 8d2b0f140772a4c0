#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_segments.py tests/test_rank_cells.py -m gpu -x -q > gpurun_out/exp6_pytest.log 2>&1
tail -15 gpurun_out/exp6_pytest.log
export KERNELS="k_cascade_tile,k_level(fused),k_smooth_vec,k_grad_mag,k_tri_x,k_triy_chns,k_resample(image)"
OUT=gpurun_out/exp6_ab.txt bash profiles/ab.sh "ACF_HIP_NO_FUSED_GRAD=1" "A=1" "ACF_HIP_NO_FUSED_GRAD=1" "A=2" "ACF_HIP_SMOOTH_SEGMENTS=8" "ACF_HIP_SMOOTH_SEGMENTS=6"
