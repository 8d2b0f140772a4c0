#!/usr/bin/env python
"""Does a HIP stream priority change how concurrently running kernels share the CUs?  Three detector contexts (bench.py's
deployment), context 0's stream created with the highest priority: per-context time of the tile and level kernels inside the
concurrent region, against the same run with equal priorities."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import capi, synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, C, B = 1080, 1920, 3, 96
model = synth.make_model(seed=1, name="FACE80")
dev = torch.device("cuda", 0)
base = torch.from_numpy(np.stack([synth.make_frame(1 + i, H, W, "luv") for i in range(2)])).to(dev)
frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device=dev)
for i in range(C * B):
    frames[i] = torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2))
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
out = {"priority_range": [lo, hi]}
for label, prios in (("equal", [0, 0, 0]), ("ctx0_high", [-1, 0, 0])):
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    dets = [HipDetector(model, H, W, 3, max_batch=B, max_hits=8192, device=0, stream=s.cuda_stream) for s in streams]
    for d in dets:
        d.set_option("scale_streams", 0)
        d.set_option("keep_pyramid", 0)
        d.set_option("cascade_turns", 5)
        d.set_option("tile_persist", 0)
        d.set_option("profile", 1)
        d.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0))

    def step():
        for i in range(C):
            with torch.cuda.stream(streams[i]):
                dets[i].run(frames[i * B:(i + 1) * B], B)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for d in dets:
        d.profile()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = []
    for d in dets:
        p = d.profile()
        per.append({k: round(p[k][0] / max(p[k][1], 1), 3) for k in ("k_cascade_tile", "k_level(fused)", "k_smooth_vec", "k_triy_chns") if k in p})
    out[label] = {"fps": round(C * B * 5 / dt), "ms_per_launch_by_context": per}
    for d in dets:
        d.close()
print(json.dumps(out, indent=1))
