# after the staged cells: frames per launch, and every scale's gradMag / x pass as thin chains with more contexts to cover them
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
echo "== 3 x 96"; run
for a in "--batch 80" "--batch 112" "--batch 128" "--opt fused_grad=2 --opt fused_tri=2"; do echo "== $a"; run $a; done
export GPU_MAX_HW_QUEUES=8
for a in "--contexts 3" "--contexts 4" "--contexts 4 --opt fused_grad=2 --opt fused_tri=2" "--contexts 5 --opt fused_grad=2 --opt fused_tri=2" "--contexts 4 --batch 80 --opt fused_grad=2 --opt fused_tri=2"; do echo "== HWQ8 $a"; run $a; done
