# with one-segment smoothing chains (smooth_segments=1 + fused_tri): every scale fused, scales on their own streams, context counts
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
B="--opt smooth_segments=1"
for a in "$B" "$B --opt fused_grad=2 --opt fused_tri=2" "$B --opt scale_streams=1" "$B --opt fused_grad=2 --opt fused_tri=2 --opt scale_streams=1" \
  "$B --contexts 4 --batch 72" "$B --contexts 4 --batch 96" "$B --contexts 2 --batch 144" "$B --contexts 6 --batch 48" "$B --turns 0" "$B --turns 1" "$B --persist 1" "$B"; do
 echo "== $a"; run $a
done
