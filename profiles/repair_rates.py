#!/usr/bin/env python
"""How often the speculative column segments miss (option count_repairs): planes recomputed as one chain per planes checked,
for the image smoothing (k_smooth_vec) and the level chains (k_level_all), against the warm-up length — 1080p synthetic frames
and frames with flat / black regions; one context, 8 frames per batch (so that both kernels segment)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acf_amd import synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, B = 1080, 1920, 8
model = synth.make_model(seed=1, name="FACE80")
fr = np.stack([synth.make_frame(100 + i, H, W, "luv") for i in range(B)])
fr[1, :, 600:900, :] = 0.0          # a black band
fr[2, :, :, 300:700] = 0.25         # a flat band
fr[3] *= 1e-3                       # a dark frame (small values: more columns until the last bit settles)
frames = torch.from_numpy(fr).cuda()
out = {}
for warm in (8, 16, 24, 32, 48, 64):
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    det.set_option("keep_pyramid", 0)
    det.set_option("count_repairs", 1)
    det.set_option("smooth_segments", 8)
    det.set_option("smooth_warm", max(16, warm // 16 * 16))
    det.set_option("level_warm", warm)
    for _ in range(3):
        det.run(frames)
    det.synchronize()
    r = det.repairs()
    out["warm_%d" % warm] = {"smooth_warm": max(16, warm // 16 * 16), "smooth_planes": r[0], "smooth_redone": r[1], "level_warm": warm, "level_planes": r[2], "level_redone": r[3]}
    det.close()
print(json.dumps(out, indent=1))
