#!/bin/bash
# A/B of the tail: old k_cascade_tail3 (ACF_HIP_TAIL3=1) vs k_tail_codes + k_tail_scan, parity of the detections, then the GPU tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_tail; rm -rf $OUT; mkdir -p $OUT
P=profiles/ubench/casc_probe.py
ACF_HIP_TAIL3=1 python $P --batch 64 --reps 5 --tag old-tail --save $OUT/ref.npz 2>&1 | tee $OUT/old.log
python $P --batch 64 --reps 5 --tag new-tail --check $OUT/ref.npz 2>&1 | tee $OUT/new.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest.log
