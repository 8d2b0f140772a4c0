# where the shared-device forms start to pay at 1080p: 3 contexts x 32 / 48 / 64 frames, a lone context's forms against the uncut chains + fused x pass
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
for b in 32 48 64; do
echo "== batch $b, lone forms"; run --batch $b --opt shared_device=0
echo "== batch $b, uncut + fused x pass"; run --batch $b --opt shared_device=0 --opt smooth_segments=1 --opt fused_tri=2
done
