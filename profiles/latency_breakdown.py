#!/usr/bin/env python
"""One 1080p frame through one context (BASELINE cfg 2 as worded): per-kernel time from the context's own HIP events
(option profile; scales in order on one stream so that spans are the kernels' own) and the submit-to-synchronise latency with the
scales on their own streams."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acf_amd import capi, synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W = 1080, 1920
model = synth.make_model(seed=1, name="FACE80")
fr = torch.from_numpy(synth.make_frame(1, H, W, "luv")[None]).cuda()
out = {}
for label, seg in (("segments_auto", 0), ("segments_off", 1)):
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=8192)
    det.set_option("keep_pyramid", 0)
    det.set_option("smooth_segments", seg)
    det.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0))
    det.set_option("scale_streams", 0)
    det.set_option("profile", 1)
    for _ in range(3):
        det.run(fr)
    det.synchronize()
    det.profile()
    for _ in range(10):
        det.run(fr)
    det.synchronize()
    prof = {k: round(1e3 * ms / n, 1) for k, (ms, n) in det.profile().items()}
    det.set_option("profile", 0)
    det.set_option("scale_streams", 1)
    lat = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        det.run(fr)
        det.synchronize()
        lat.append(time.perf_counter() - t0)
    out[label] = {"latency_ms": 1e3 * float(np.median(lat[2:])), "us_per_launch_in_order": prof}
    det.close()
print(json.dumps(out, indent=1))
