#!/bin/bash
# Round-2 baseline of the cascade alone on a resident pyramid: tiled path, staged (no-LDS) path, stage stamps, PMC of both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_probe; rm -rf $OUT; mkdir -p $OUT
cd $R
P=profiles/ubench/casc_probe.py
python $P --batch 64 --reps 5 --tag tiled --save $OUT/ref.npz --full 2>&1 | tee $OUT/tiled.log
python $P --batch 64 --reps 5 --tag staged --opt cascade_tiles=0 --check $OUT/ref.npz 2>&1 | tee $OUT/staged.log
ACF_HIP_CASC_DEBUG=4 python $P --batch 64 --reps 2 --tag stamps 2>&1 | grep -i "stamps\|==" | tee $OUT/stamps.log
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  T=$(echo $SET | cut -c1-12 | tr ' ' _)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_$T -o pmc --output-format csv -- python $P --batch 64 --reps 2 --tag pmc > $OUT/pmc_$T.log 2>&1
  F=$(find $OUT/pmc_$T -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python profiles/pmc_dispatches.py "$F" | grep -i "kernel\|casc" | tee $OUT/pmc_$T.csv
  rm -rf $OUT/pmc_$T
done
