import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from acf_amd import synth
from acf_amd.detector import HipDetector
from oracle import binding as ob
for (H, W) in [(1200, 128), (1088, 128), (1040, 128), (1024, 128)]:
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    frame = synth.make_frame(23, H, W, "luv")
    det = HipDetector(model, H, W, 3, max_batch=2, max_hits=1 << 15)
    det.set_option("fused_smooth", 1); det.set_option("fused_tri", 2); det.set_option("fused_grad", 2); det.set_option("scale_streams", int(os.environ.get("SS", "1")))
    det.run(torch.from_numpy(np.stack([frame, frame])).cuda())
    plan = ob.Plan(model, H, W, 3)
    pyr, _, _ = ob.chns_pyramid(plan, frame)
    got = det.read_pyramid(0)
    L = plan.levels[0]
    n0 = 10 * L.hP * L.wP
    a = got[:n0].view(np.uint32).reshape(10, L.wP, L.hP); b = pyr[:n0].view(np.uint32).reshape(10, L.wP, L.hP)
    bad = np.argwhere(a != b)
    print(H, W, "level 0:", L.hP, L.wP, "bad cells", len(bad), "channels", sorted(set(bad[:, 0])) if len(bad) else None, "rows", (bad[:, 2].min(), bad[:, 2].max()) if len(bad) else None, "cols", (bad[:, 1].min(), bad[:, 1].max()) if len(bad) else None, "total bad", int((got.view(np.uint32) != pyr.view(np.uint32)).sum()))
    det.close()
