#!/bin/bash
# A/B of the rank-cell tile kernel's geometry (rows of windows per tile x waves per tile): k_cascade_tile ms per 96-frame
# launch alone on the GPU (bench.py's roofline.solo), written to gpurun_out/r03_rtile_ab.txt
out=gpurun_out/r03_rtile_ab.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['solo']['kernels_ms_per_launch']
print(round(d['value']), {k:s[k] for k in ('k_cascade_tile','k_rank','k_tail_scan') if k in s})" >> $out 2>&1
}
run ACF_HIP_NO_RANK=1
run ACF_HIP_RTILE_TR=32 ACF_HIP_RTILE_NW=8
run ACF_HIP_RTILE_TR=32 ACF_HIP_RTILE_NW=4
run ACF_HIP_RTILE_TR=64 ACF_HIP_RTILE_NW=8
run ACF_HIP_RTILE_TR=16 ACF_HIP_RTILE_NW=8
run ACF_HIP_RTILE_TR=16 ACF_HIP_RTILE_NW=4
run ACF_HIP_RTILE_TR=64 ACF_HIP_RTILE_NW=4
run ACF_HIP_RTILE_TR=32 ACF_HIP_RTILE_NW=8 ACF_HIP_TILE_TB8=1
run ACF_HIP_RTILE_TR=32 ACF_HIP_RTILE_NW=8 ACF_HIP_CASC_BOUNDS=16,32,64,128
run ACF_HIP_RTILE_TR=32 ACF_HIP_RTILE_NW=8 ACF_HIP_CASC_BOUNDS=24,24,64,128
cat $out
