#!/bin/bash
cd $GRAFT_REPO_ROOT
export KERNELS="k_grad_mag,k_smooth_vec,k_triy_chns,k_tri_x"
OUT=gpurun_out/exp4_ab.txt bash profiles/ab.sh "A=1" "ACF_HIP_GMV_BLOCKS=512" "A=2" "ACF_HIP_GMV_BLOCKS=512" "ACF_HIP_GMV_BLOCKS=384"
