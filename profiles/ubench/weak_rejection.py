#!/usr/bin/env python
"""The pooled tile kernel when early rejection is weak (a low cascThr keeps many windows alive past tree 32): k_cascade_tile3
against round 3.s k_cascade_tile2 (deleted in round 5: the figures in DESIGN.md 3.0 were taken with ACF_HIP_TILE2=1 on the round-4 library), us per 1080p frame of cascade, over a sweep of cascThr."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, B = 1080, 1920, 24
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(2)])).cuda()
frames = torch.stack([torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2)) for i in range(B)])
out = {}
for thr in (-1.0, -2.0, -3.0, -5.0):
    model = synth.make_model(seed=1, name="FACE80", cascThr=thr)
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=1 << 18)
    det.set_option("keep_pyramid", 0)
    det.set_option("profile", 1)
    for _ in range(2):
        det.run(frames)
    det.synchronize()
    det.profile()
    for _ in range(3):
        det.run(frames)
    det.synchronize()
    prof = det.profile()
    out["cascThr_%g" % thr] = {k: round(1e3 * ms / (3 * B), 1) for k, (ms, n) in prof.items() if "casc" in k or "tail" in k}
    det.close()
print(json.dumps(out))
