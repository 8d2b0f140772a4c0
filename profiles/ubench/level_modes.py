#!/usr/bin/env python
"""k_level_all by what it writes: floats only (rank_cells = 0), rank cells only (the detection-only call:
keep_pyramid = 0), both (keep_pyramid = 1) — ms per 96 frames of 1080p, one context, scales in order on one stream."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, B = 1080, 1920, 96
model = synth.make_model(seed=1, name="FACE80")
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(2)])).cuda()
frames = torch.stack([torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2)) for i in range(B)])
out = {}
for name, keep, rank in (("floats_only", 1, 0), ("ranks_only", 0, 1), ("both", 1, 1)):
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    det.set_option("scale_streams", 0)
    det.set_option("keep_pyramid", keep)
    det.set_option("rank_cells", rank)
    det.set_option("profile", 1)
    for _ in range(2):
        det.run(frames)
    det.synchronize()
    det.profile()
    for _ in range(4):
        det.run(frames)
    det.synchronize()
    p = det.profile()
    out[name] = {k: round(p[k][0] / p[k][1], 4) for k in ("k_level(fused)", "k_cascade_tile", "k_rank") if k in p}
    det.close()
print(json.dumps(out, indent=1))
