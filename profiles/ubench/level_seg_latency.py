#!/usr/bin/env python
"""Single-frame latency (cfg 2 as worded) against the level kernel's column segments: option level_segments (0 = auto) and
level_warm over several synthetic frames — a segment whose warm-up was too short is repaired by a second, whole-chain launch,
so the median AND the worst frame are reported, with the number of planes repaired."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import capi, synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W = 1080, 1920
model = synth.make_model(seed=1, name="FACE80")
frames = [torch.from_numpy(synth.make_frame(100 + i, H, W, "luv")[None]).cuda() for i in range(6)]
out = {}
for seg, warm in ((1, 32), (0, 32), (0, 48), (0, 64), (0, 96)):
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=8192)
    det.set_option("keep_pyramid", 0)
    det.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0))
    det.set_option("level_segments", seg)
    det.set_option("level_warm", warm)
    det.set_option("count_repairs", 1)
    per = []
    rep = []
    for fr in frames:
        for _ in range(3):
            det.run(fr)
        det.synchronize()
        r0 = det.repairs()
        lat = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            det.run(fr)
            det.synchronize()
            lat.append(time.perf_counter() - t0)
        r1 = det.repairs()
        per.append(1e3 * float(np.median(lat)))
        rep.append((r1[3] - r0[3]) // 8)
    out["segments_%d_warm_%d" % (seg, warm)] = {"latency_ms_per_frame": [round(x, 3) for x in per], "median": round(float(np.median(per)), 3),
                                                 "worst": round(max(per), 3), "level_planes_repaired_per_run": rep}
    det.close()
print(json.dumps(out, indent=1))
