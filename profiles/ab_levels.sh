# A/B of the level stage: fused_levels 0 (separate launches), 2 (separate resample + wave-per-plane smoothing), 1 (fused)
for M in 0 2 1; do
python - <<PY
import json, subprocess, sys, os
os.environ["ACF_BENCH_LEVEL_MODE"] = "$M"
out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print("mode $M", round(d["value"]), d["roofline"]["kernels_ms_per_step"])
PY
done
