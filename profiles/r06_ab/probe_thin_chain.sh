#!/bin/bash
# thin-chain sensitivity: duration (s_sleep) and work (idle waves), 3 x 96 frames, one box
export ACF_HIP_LIB=$GRAFT_REPO_ROOT/acf_amd/libacf_hip_probe.so
out=gpurun_out/probe.txt; : > $out
run() { echo "== $1" >> $out; env $1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['value']), d['value_repeats'], 'tri solo ms', r['solo']['kernels_ms_per_batch'].get('k_smooth_grad_tri'), 'in-region', r['kernels_ms_per_step'].get('k_smooth_grad_tri'))" >> $out 2>&1; }
run "X=0"
run "ACF_PROBE_SLEEP=3"
run "ACF_PROBE_SLEEP=6"
run "ACF_PROBE_SLEEP=12"
run "ACF_PROBE_WAVES=1"
run "ACF_PROBE_WAVES=3"
run "X=1"
cat $out
