"""Feasibility probe: does the cascade of one batch overlap with the pyramid of another on two streams?
Two contexts with their own streams; times pyramid alone, cascade alone, both issued back to back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from acf_amd import synth
from acf_amd.detector import HipDetector
H, W, B = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synth.make_model(seed=1, name="FACE80")
base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
frames = torch.stack([torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2)) for i in range(B)])
A = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
Bd = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
A.run(frames); Bd.run(frames); A.synchronize(); Bd.synchronize()
def t(fn, n=5):
    fn(); A.synchronize(); Bd.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    A.synchronize(); Bd.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
pa = t(lambda: A.pyramid(frames))
cb = t(lambda: Bd.detect())
both = t(lambda: (A.pyramid(frames), Bd.detect()))
print("batch %d: pyramid %.2f ms, cascade %.2f ms, both concurrently %.2f ms (sum %.2f, max %.2f)" % (B, pa, cb, both, pa + cb, max(pa, cb)))
