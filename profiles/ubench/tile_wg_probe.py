#!/usr/bin/env python
"""How does the tile kernel's time depend on the tiles a CU holds?  A 7-channel model (colour disabled: 70 % of the cell bytes) on 1080p
frames lets four 8-wave tile workgroups fit a CU's LDS (ACF_HIP_RTILE_WG=4) — if the kernel also stays within 64 VGPRs (a library built with
-DACF_TILE3_OCC='__attribute__((amdgpu_waves_per_eu(8,8)))', selected with ACF_HIP_LIB).  Prints the tile kernel's ms per 96 frames."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import capi, synth
from acf_amd.detector import HipDetector
H, W, B = 1080, 1920, 96
model = synth.make_model(seed=1, name="FACE80", colorEnabled=0)
dev = torch.device("cuda", 0)
base = torch.from_numpy(np.stack([synth.make_frame(1 + i, H, W, "luv") for i in range(2)])).to(dev)
frames = torch.empty((B, 3, W, H), dtype=torch.float32, device=dev)
for i in range(B):
    frames[i] = torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2))
det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
det.set_option("scale_streams", 0)
det.set_option("keep_pyramid", 0)
det.set_option("tile_persist", int(os.environ.get("PERSIST", "0")))
det.set_option("profile", 1)
for _ in range(2):
    det.run(frames, B)
det.synchronize(); det.profile()
for _ in range(5):
    det.run(frames, B)
det.synchronize()
p = det.profile()
print(json.dumps({"env": {k: os.environ.get(k) for k in ("ACF_HIP_LIB", "ACF_HIP_RTILE_WG", "PERSIST")}, "tile_ms": round(p["k_cascade_tile"][0] / 5, 4),
                  "level_ms": round(p["k_level(fused)"][0] / 5, 4), "nChns": det.nChns}))
