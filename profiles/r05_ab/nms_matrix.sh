python -m pytest tests -m gpu -x -q -k "nms" 2>&1 | tail -2
python profiles/latency_breakdown.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k:v for k,v in d['segments_auto']['us_per_launch_in_order'].items() if k in ('k_nms','k_sort_map')}, d['segments_auto']['latency_ms'])"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-repeats 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['solo']['kernels_ms_per_launch']; print(round(d['value']), 'nms solo ms', s.get('k_nms'), 'latency', round(d['latency_ms_batch1'],4), 'batch8', round(d['config']['batch8_fps_1gpu']), round(d['config']['batch8_pipelined_fps_1gpu']), d['verified_frames'])"
