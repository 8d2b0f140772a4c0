#!/bin/bash
# round-3 session-3 experiment: VALU issue rates incl. packed f32, parity of the level-kernel changes, A/B of the SLP vectoriser
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./profiles/ubench/valu_rate > gpurun_out/valu_rate.txt 2>&1
timeout 900 python -m pytest tests/test_rank_cells.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/exp1_pytest.log 2>&1
tail -3 gpurun_out/exp1_pytest.log
export KERNELS="k_cascade_tile,k_level(fused),k_smooth_vec,k_grad_mag,k_tri_x,k_triy_chns,k_resample(image),k_tail_scan"
OUT=gpurun_out/exp1_ab.txt bash profiles/ab.sh "A=1" "ACF_HIP_LIB=acf_amd/libacf_hip_noslp.so" "A=2" "ACF_HIP_LIB=acf_amd/libacf_hip_noslp.so"
tail -12 gpurun_out/valu_rate.txt
