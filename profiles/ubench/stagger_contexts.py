"""Does a deliberate phase offset between detector contexts help?  Context i+1's first batch starts when context i's
first pyramid is done, so pyramids (HBM-bound) and cascades (VALU/LDS-bound) of different contexts interleave from the start."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from acf_amd import synth
from acf_amd.detector import DetectorPool
H, W = 1080, 1920
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
model = synth.make_model(seed=1, name="FACE80")
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device="cuda")
for i in range(C * B):
    frames[i] = torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2))
pool = DetectorPool(C, model, H, W, 3, max_batch=B, max_hits=8192)
cap = 1024
recs = [torch.zeros((B, 1 + 6 * cap), dtype=torch.int32, device="cuda") for _ in range(C)]


def run(stagger):
    for k in range(steps):
        prev_ev = None
        for i, (d, s) in enumerate(pool):
            with torch.cuda.stream(s):
                if stagger and k == 0:
                    if prev_ev is not None:
                        s.wait_event(prev_ev)
                    d.pyramid(frames[i * B:(i + 1) * B], B)
                    prev_ev = torch.cuda.Event()
                    prev_ev.record(s)
                    d.detect()
                else:
                    d.run(frames[i * B:(i + 1) * B], B)
                d.export_detections(recs[i], cap)
    pool.synchronize()
    torch.cuda.synchronize()


for stagger in (0, 1, 0, 1):
    run(0)
    t0 = time.perf_counter()
    run(stagger)
    dt = time.perf_counter() - t0
    print("stagger %d: %d x %d, %d steps: %.0f FPS" % (stagger, C, B, steps, C * B * steps / dt))
