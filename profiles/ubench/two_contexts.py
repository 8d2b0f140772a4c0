"""Throughput probe: two detector contexts on their own streams, batches alternating between them, so that the
cascade of batch k (VALU/LDS-bound) overlaps the pyramid of batch k+1 (HBM-bound).  Compared with one context."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from acf_amd import synth
from acf_amd.detector import HipDetector
H, W = 1080, 1920
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = synth.make_model(seed=1, name="FACE80")
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
frames = torch.empty((B, 3, W, H), dtype=torch.float32, device="cuda")
for i in range(B):
    frames[i] = torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2))
torch.cuda.synchronize()
cap = 1024
NC = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2]
dets = [HipDetector(model, H, W, 3, max_batch=B, max_hits=8192) for _ in range(max(NC))]
recs = [torch.zeros((B, 1 + 6 * cap), dtype=torch.int32, device="cuda") for _ in range(max(NC))]


def run(n_ctx, steps):
    for k in range(steps):
        d = dets[k % n_ctx]
        d.run(frames, B)
        d.export_detections(recs[k % n_ctx], cap)
    for d in dets:
        d.synchronize()


for n_ctx in NC:
    run(n_ctx, 2)
    t0 = time.perf_counter()
    steps = 12
    run(n_ctx, steps)
    dt = time.perf_counter() - t0
    print("contexts %d batch %d: %.0f FPS (%.2f ms per batch)" % (n_ctx, B, B * steps / dt, dt / steps * 1e3))
