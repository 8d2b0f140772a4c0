"""The deployment bench.py times, as a test: three detector contexts on their own streams (acf_amd.detector.DetectorPool: turns for
the VALU-bound kernels, one tile workgroup per tile), batches of 1080p frames resident in HBM, the detection-only call
(keep_pyramid = 0: the levels leave as 16-bit rank cells), the device bbNms + prune, the fixed-capacity record export — run
concurrently for several steps, then six sampled frames of the LAST step checked against the oracle (chnsPyramid + acfDetect +
bbNms + prune on the CPU): count, boxes, levels, score bits.  bench.py's own self-check does the same after the clock stops;
this keeps the configuration green in the test suite."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_three_contexts_detection_only_with_device_nms():
    import torch
    import bench
    from acf_amd import capi, synth
    from acf_amd.detector import DetectorPool
    from acf_amd.dist import RecordGather
    H, W, C, B, cap = 1080, 1920, 3, 16, 32
    model = synth.make_model(seed=1, name="FACE80")
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(np.stack([synth.make_frame(4000 + i, H, W, "luv") for i in range(3)])).to(dev)
    frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device=dev)
    for i in range(C * B):
        frames[i] = torch.roll(base[i % 3], shifts=(41 * (i // 3), 29 * (i // 3)), dims=(1, 2))
    pool = DetectorPool(C, model, H, W, 3, max_batch=B, max_hits=8192, device=0)
    nms = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)
    for det in pool.dets:
        det.set_option("scale_streams", 0)
        det.set_option("keep_pyramid", 0)
        det.set_nms(nms)
    pipes = [RecordGather(B, 1 + 6 * cap, 1, 0, dev) for _ in range(C)]
    for _ in range(3):
        for i in range(C):
            with torch.cuda.stream(pool.streams[i]):
                rec = pipes[i].buffer()
                pool.dets[i].run(frames[i * B:(i + 1) * B], B)
                pool.dets[i].export_detections(rec, cap)
                pipes[i].submit()
    for i in range(C):
        with torch.cuda.stream(pool.streams[i]):
            pipes[i].finish()
    torch.cuda.synchronize()
    recs = [pipes[i].rec[(pipes[i].k - 1) & 1].cpu().numpy() for i in range(C)]
    picks = [(i % C, (3 + 5 * i) % B) for i in range(6)]
    n = bench.verify_frames(model, H, W, [frames[i * B:(i + 1) * B] for i in range(C)], recs, cap, nms, picks)
    assert n == 6
    assert sum(int(r[:, 0].sum()) for r in recs) > 0
    pool.close()
