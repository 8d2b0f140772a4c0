#!/bin/bash
# A/B of the working tree's library against acf_amd/libacf_hip_base.so (+ parity of the working tree)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rank_cells.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_segments.py -m gpu -x -q > gpurun_out/exp3_pytest.log 2>&1
tail -3 gpurun_out/exp3_pytest.log
export KERNELS="k_cascade_tile,k_level(fused),k_smooth_vec,k_grad_mag,k_tri_x,k_triy_chns,k_resample(image),k_tail_scan,k_nms"
OUT=gpurun_out/exp3_ab.txt bash profiles/ab.sh "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=1" "ACF_HIP_LIB=acf_amd/libacf_hip_base.so" "A=2"
