"""One 1080p frame through one context: submit -> synchronise, plain launches against the captured graph (option graph)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from acf_amd import synth, capi
from acf_amd.detector import HipDetector
H, W = 1080, 1920
model = synth.make_model(seed=1, name="FACE80")
frames = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(2)])).cuda()
for ss in (1, 0):
    for graph in (0, 1):
        det = HipDetector(model, H, W, 3, max_batch=1, max_hits=8192)
        det.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0))
        det.set_option("keep_pyramid", 0)
        det.set_option("scale_streams", ss)
        det.set_option("graph", graph)
        lat = []
        res = []
        for k in range(14):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            det.run(frames[k % 2 if k >= 10 else 0:][:1], 1)   # the last calls alternate the input: re-capture each time
            det.synchronize()
            lat.append(time.perf_counter() - t1)
            if k in (0, 3, 9, 12, 13):
                res.append(det.detections(0)[0].tobytes())
        assert res[0] == res[1] == res[2] == res[3], "graph replay differs"
        print("scale_streams", ss, "graph", graph, "median ms", round(1e3 * float(np.median(lat[3:10])), 3), "recapture ms", round(1e3 * lat[11], 3), "same frame results equal:", res[0] == res[3], "other frame differs:", res[3] != res[4])
        det.close()
