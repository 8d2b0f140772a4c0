# tile_persist = n persistent tile workgroups (fewer than fill the machine: room for the other contexts' kernels) against one workgroup per tile
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
for p in 0 256 384 512 640 0; do echo "== --persist $p"; run --persist $p; done
