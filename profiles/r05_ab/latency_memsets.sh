python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python profiles/latency_breakdown.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['segments_auto'])"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['value_repeats'], 'latency', round(d['latency_ms_batch1'],4), 'batch8', round(d['config']['batch8_fps_1gpu']), round(d['config']['batch8_pipelined_fps_1gpu']), d['verified_frames'])"
