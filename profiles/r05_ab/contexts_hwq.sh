# more contexts need more hardware queues (the runtime's default is 4): 3..6 contexts x 96 / 72 / 64 frames with GPU_MAX_HW_QUEUES=8
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value_repeats'))"; }
echo "== default queues, 3 x 96"; run
export GPU_MAX_HW_QUEUES=8
for a in "--contexts 3 --batch 96" "--contexts 4 --batch 96" "--contexts 5 --batch 96" "--contexts 6 --batch 96" "--contexts 4 --batch 72" "--contexts 6 --batch 64" "--contexts 4 --batch 128"; do
 echo "== HWQ 8 $a"; run $a
done
