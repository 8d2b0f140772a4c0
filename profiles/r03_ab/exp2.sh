#!/bin/bash
# e32 -> e64 v_cndmask peephole: parity + A/B
cd $GRAFT_REPO_ROOT
ACF_HIP_LIB=acf_amd/libacf_hip_e64.so timeout 900 python -m pytest tests/test_rank_cells.py tests/test_gpu_pipeline.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/exp2_pytest.log 2>&1
tail -3 gpurun_out/exp2_pytest.log
export KERNELS="k_cascade_tile,k_level(fused),k_smooth_vec,k_grad_mag,k_tri_x,k_triy_chns,k_resample(image),k_tail_scan,k_nms"
OUT=gpurun_out/exp2_ab.txt bash profiles/ab.sh "A=1" "ACF_HIP_LIB=acf_amd/libacf_hip_e64.so" "A=2" "ACF_HIP_LIB=acf_amd/libacf_hip_e64.so"
