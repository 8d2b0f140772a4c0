"""PCIe-inclusive rates of the host-buffer entries (never bench.py's `value`; DESIGN.md §4 quotes them).

  python profiles/stream_bench.py [--batch 64] [--steps 6]

Prints one JSON line: frames/s for
  hbm_u8      packed 8-bit RGB frames resident in HBM -> acf_hip_run_u8 (ingest + rgb2luv fused)
  host_f32    acf_hip_run_host: planar f32 frames in pageable host memory, blocking
  stream_u8   acf_hip_stream_submit/collect: packed 8-bit RGB frames in pinned host memory, depth 2 and 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    import torch
    from acf_amd import capi, synth
    from acf_amd.detector import HipDetector, PinnedBuffer
    H, W, B = args.height, args.width, args.batch
    model = synth.make_model(seed=1, name="FACE80", isLuv=0)  # RGB in, rgb2luv on the device
    rgb = synth.make_frame(7, H, W, "rgb")
    up = np.clip(np.rint(rgb.transpose(2, 1, 0) * 255.0), 0, 255).astype(np.uint8)  # [H][W][3]
    batch = np.ascontiguousarray(np.stack([np.roll(np.ascontiguousarray(up), (13 * i, 29 * i), axis=(0, 1)) for i in range(B)]))
    out = {"batch": B, "frame": "%dx%d" % (W, H)}

    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    x = torch.from_numpy(batch).cuda()
    for _ in range(2):
        det.run_u8(x, capi.PIX_RGB)
    det.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det.run_u8(x, capi.PIX_RGB)
    det.synchronize()
    out["hbm_u8_fps"] = B * args.steps / (time.perf_counter() - t0)
    del x

    for depth in (2, 3):
        pins = [PinnedBuffer(batch.nbytes) for _ in range(depth)]
        for p in pins:
            p.array[:] = batch.ravel()
        det.stream_open(capi.PIX_RGB, 0, 1024, depth)
        tickets = []
        for k in range(depth):
            tickets.append(det.stream_submit(pins[k].ptr.value, B))
        while tickets:
            det.stream_collect(tickets.pop(0))
        t0 = time.perf_counter()
        for k in range(args.steps):
            if len(tickets) == depth:
                det.stream_collect(tickets.pop(0))
            tickets.append(det.stream_submit(pins[k % depth].ptr.value, B))
        while tickets:
            det.stream_collect(tickets.pop(0))
        out["stream_u8_depth%d_fps" % depth] = B * args.steps / (time.perf_counter() - t0)
        det.stream_close()
        for p in pins:
            p.close()
    det.close()

    model = synth.make_model(seed=1, name="FACE80")
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    luv = synth.make_frame(7, H, W, "luv")
    fb = np.ascontiguousarray(np.broadcast_to(luv, (B,) + luv.shape))
    det.run_host(fb)
    det.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(2, args.steps // 2)):
        det.run_host(fb)
    det.synchronize()
    out["host_f32_fps"] = B * max(2, args.steps // 2) / (time.perf_counter() - t0)
    det.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
