import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from acf_amd import synth
from acf_amd.detector import HipDetector
H, W, B = 1080, 1920, 48
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
frames = torch.stack([torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2)) for i in range(B)])
for depth in (1, 3, 4):
    model = synth.make_model(seed=1, name="FACE80", treeDepth=depth)
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    det.set_option("profile", 1)
    for _ in range(2):
        det.run(frames)
    det.synchronize(); det.profile()
    det.run(frames); det.synchronize()
    prof = det.profile()
    print(depth, {k: round(1e3*ms/B,1) for k,(ms,n) in prof.items() if 'casc' in k or 'tail' in k})
    det.close()
