# the tile kernel's shape / stage boundaries and a few others, re-swept with the shared-device forms (uncut smoothing chains, fused x pass)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
echo "== default"; run
for e in "ACF_HIP_RTILE_TR=16" "ACF_HIP_RTILE_TR=64" "ACF_HIP_RTILE_NW=4" "ACF_HIP_RTILE_WG=2" "ACF_HIP_RTILE_WG=4" "ACF_HIP_CASC_BOUNDS=8,32,32,128" "ACF_HIP_CASC_BOUNDS=12,32,32,128" \
  "ACF_HIP_CASC_BOUNDS=16,48,48,128" "ACF_HIP_CASC_BOUNDS=16,32,32,96" "ACF_HIP_CASC_BOUNDS=16,32,32,192" "ACF_HIP_CASC_BOUNDS=24,48,48,128" "ACF_HIP_GMV_BLOCKS=128" "ACF_HIP_GMV_BLOCKS=512" "ACF_HIP_TILE_PAD_KB=8"; do
 echo "== $e"; env $e bash -c "$(declare -f run); run"
done
echo "== default"; run
