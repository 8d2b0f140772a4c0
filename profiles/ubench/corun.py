#!/usr/bin/env python
"""How complementary are the two halves of the path?  Context A loops acf_hip_pyramid (smoothing, gradMag, running sums, channel
cells, resamples, level kernel: mostly memory-bound), context B loops acf_hip_detect on a pyramid it computed once (tile kernel,
tail scan, sort, NMS: latency / VALU-bound).  Rates alone and side by side; 1.0 / 1.0 side by side would be perfect overlap,
0.5 / 0.5 none."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import capi, synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

H, W, B = 1080, 1920, 96
model = synth.make_model(seed=1, name="FACE80")
dev = torch.device("cuda", 0)
base = torch.from_numpy(np.stack([synth.make_frame(1 + i, H, W, "luv") for i in range(2)])).to(dev)
frames = torch.empty((B, 3, W, H), dtype=torch.float32, device=dev)
for i in range(B):
    frames[i] = torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2))
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
dets = [HipDetector(model, H, W, 3, max_batch=B, max_hits=8192, device=0, stream=s.cuda_stream) for s in streams]
for d in dets:
    d.set_option("scale_streams", 0)
    d.set_option("keep_pyramid", 0)
    d.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0))
persist = int(os.environ.get("PERSIST", "0"))
dets[1].set_option("tile_persist", persist)
dets[1].run(frames, B)
dets[1].synchronize()


def loop(i, fn, n):
    with torch.cuda.stream(streams[i]):
        for _ in range(n):
            fn()


def rate(which, n=12, reps=1):
    fns = {0: lambda: dets[0].pyramid(frames, B), 1: lambda: dets[1].detect()}
    for w in which:
        loop(w, fns[w], 2)
    torch.cuda.synchronize()
    ev = {}
    for w in which:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[w])
        loop(w, fns[w], n if w == 0 else n * reps)
        e1.record(streams[w])
        ev[w] = (e0, e1)
    torch.cuda.synchronize()
    return {("pyramid" if w == 0 else "detect"): round(ev[w][0].elapsed_time(ev[w][1]) / (n if w == 0 else n * reps), 3) for w in which}


out = {"tile_persist": persist, "alone_ms_per_batch": {**rate([0]), **rate([1])}}
a = out["alone_ms_per_batch"]
# side by side: detect gets enough iterations to stay busy for about as long as the pyramid loop does (so both rates are steady-state ones)
out["together_ms_per_batch"] = rate([0, 1], 12, max(1, int(round(a["pyramid"] / a["detect"]))))
t = out["together_ms_per_batch"]
out["throughput_sum_vs_alone"] = round(a["pyramid"] / t["pyramid"] + a["detect"] / t["detect"], 3)  # 1 = no gain from overlap, 2 = perfect
print(json.dumps(out))
for d in dets:
    d.close()
