#!/bin/bash
cd $GRAFT_REPO_ROOT
ACF_HIP_GMV_GLOBAL_PX=100000000 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
export KERNELS="k_grad_mag,k_smooth_vec,k_triy_chns"
OUT=gpurun_out/exp13_ab.txt bash profiles/ab.sh "A=1" "ACF_HIP_GMV_GLOBAL_PX=600000" "A=2" "ACF_HIP_GMV_GLOBAL_PX=600000" "ACF_HIP_GMV_GLOBAL_PX=200000"
