#!/bin/bash
# co-scheduling knobs in the 3-context bench (frames/s, ms per step)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp5_args.txt bash profiles/ab_args.sh "" "--no-profile" "--contexts 4 --no-profile" "--contexts 4 --batch 72 --no-profile" "--contexts 2 --batch 144 --no-profile" "--contexts 6 --batch 48 --no-profile" "--no-profile"
for e in "ACF_HIP_TILE_PAD_KB=10" "ACF_HIP_CASCADE_TURNS=1" "ACF_HIP_CASCADE_TURNS=7" "ACF_HIP_CASCADE_TURNS=0" "ACF_HIP_GMV_BLOCKS=512" "A=1"; do
  echo "== $e" >> gpurun_out/exp5_args.txt
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-verify --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'])" >> gpurun_out/exp5_args.txt 2>&1
done
cat gpurun_out/exp5_args.txt
