# A/B of the tiled cascade's stage boundaries (tuning knob ACF_HIP_CASC_BOUNDS="b1,b2,b3,b4")
for B in "$@"; do
 export ACF_HIP_CASC_BOUNDS=$B
 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$B', round(d['value']), d['roofline']['kernels_ms_per_step']['k_cascade'])"
done
