#!/usr/bin/env python
"""What a model pays for not being depth 2: the LDS-tiled cascade (k_cascade_tile2) serves depth-2 trees
only; every other depth runs the global-memory staged cascade (k_cascade_first / _queue / _tail).  Same 1080p workload as
bench.py, one context, FACE80-shaped models of depth 1, 2, 3 (and depth 2 with the tile path switched off): us per frame of the
cascade kernels and of the whole path."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acf_amd import synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, B = 1080, 1920, 48
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
frames = torch.stack([torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2)) for i in range(B)])
out = {}
for name, depth, tiles in (("depth1", 1, 1), ("depth1_staged", 1, 0), ("depth2", 2, 1), ("depth2_staged", 2, 0), ("depth3", 3, 1), ("depth3_staged", 3, 0),
                          ("depth4", 4, 1), ("depth4_staged", 4, 0)):
    model = synth.make_model(seed=1, name="FACE80", treeDepth=depth)
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    det.set_option("cascade_tiles", tiles)
    det.set_option("profile", 1)
    for _ in range(2):
        det.run(frames)
    det.synchronize()
    det.profile()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        det.run(frames)
    ev1.record()
    det.synchronize()
    prof = det.profile()
    casc = sum(ms for k, (ms, n) in prof.items() if k.startswith("k_cascade") or k in ("k_tail_scan", "k_rank"))
    out[name] = {"us_per_frame_path": 1e3 * ev0.elapsed_time(ev1) / (3 * B), "us_per_frame_cascade": 1e3 * casc / (3 * B),
                 "mean_detections": float(np.mean([len(det.detections(f)[0]) for f in range(4)]))}
    det.close()
print(json.dumps(out, indent=1))
