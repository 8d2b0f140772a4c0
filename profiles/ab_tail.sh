#!/bin/bash
# A/B of the cascade tail kernel phases (ACF_HIP_CASC_DEBUG bits 16: skip phase 2, 32: skip the footprint fill, 64: skip the leaf loop)
for d in 0 16 48 80 112; do
  echo "debug=$d"
  ACF_HIP_CASC_DEBUG=$d python bench.py --batch ${1:-64} --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernels_ms_per_step']['k_cascade_tail2'])"
done
echo tail2; ACF_HIP_TAIL2=1 python bench.py --batch ${1:-64} --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernels_ms_per_step']['k_cascade_tail2'])"
