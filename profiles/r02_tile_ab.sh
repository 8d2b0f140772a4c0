#!/bin/bash
# A/B of k_cascade_tile2 variants on a resident pyramid (cascade alone): REF = the committed build's hits are not available on the
# box, so the first run saves its hits and every variant is checked against it; the GPU tests (oracle parity) run at the end.
# usage: VARIANTS="ENV=val;ENV2=val ..." bash profiles/r02_tile_ab.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_tile; rm -rf $OUT; mkdir -p $OUT
P=profiles/ubench/casc_probe.py
python $P --batch 96 --reps 5 --tag default --save $OUT/ref.npz 2>&1 | grep "==\|k_cascade_tile \|k_tail" | tee $OUT/ab.log
ACF_HIP_CASC_DEBUG=4 python $P --batch 96 --reps 2 --tag "stamps" 2>&1 | grep "casc stamps" | tail -1 | tee -a $OUT/ab.log
for V in $VARIANTS; do
  env $(echo $V | tr ";" " ") python $P --batch 96 --reps 5 --tag "$V" --check $OUT/ref.npz 2>&1 | grep "==\|k_cascade_tile \|parity" | tee -a $OUT/ab.log
done
python bench.py --no-latency 2>/dev/null | grep '^{"metric"' > $OUT/bench.json; python -c "import json; d=json.load(open('$OUT/bench.json')); print('bench default', d['value'])" | tee -a $OUT/ab.log
[ -n "$NOTEST" ] || timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest.log
