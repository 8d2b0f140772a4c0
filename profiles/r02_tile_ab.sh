#!/bin/bash
# A/B of the tile kernel: round 1's k_cascade_tile + k_cascade_tail3 vs k_cascade_tile2 (+ in-tile tail codes) + k_tail_scan
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_tile; rm -rf $OUT; mkdir -p $OUT
P=profiles/ubench/casc_probe.py
ACF_HIP_TILE1=1 ACF_HIP_TAIL3=1 python $P --batch 64 --reps 5 --tag r1-kernels --save $OUT/ref.npz 2>&1 | tee $OUT/old.log
python $P --batch 64 --reps 5 --tag tile2 --check $OUT/ref.npz $EXTRA 2>&1 | tee $OUT/new.log
for B in $BOUNDS; do
  ACF_HIP_CASC_BOUNDS=$B python $P --batch 64 --reps 5 --tag "tile2 bounds $B" --check $OUT/ref.npz 2>&1 | tee -a $OUT/bounds.log
done
[ -n "$NOTEST" ] || timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest.log
