cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python profiles/ubench/casc_probe.py --batch 64 --reps 5 --tag cleanup --full 2>&1 | grep -v amdgpu
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FPS',d['value'],'ms/step',d['ms_per_step'], d['config']['mean_detections_per_frame'], d['config']['gather_record_bytes_per_frame'])"
