#!/usr/bin/env python
"""Cascade-only timing on a resident 1080p FACE80 pyramid: the pyramid is built once, then acf_hip_detect is
repeated and every kernel's HIP-event time is printed (us per frame).  Variants are selected with the library's
ACF_HIP_* environment knobs and --opt key=value (acf_hip_set_option); --check compares the hits of every frame
with a reference run of the default configuration written by --save.

    python profiles/ubench/casc_probe.py --batch 64 --reps 5 [--opt cascade_tiles=0] [--save f.npz | --check f.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--model", default="FACE80")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--save")
    ap.add_argument("--check")
    ap.add_argument("--tag", default="")
    ap.add_argument("--full", action="store_true", help="time the whole path (pyramid + detect) as well")
    args = ap.parse_args()
    import torch
    from acf_amd import synth
    from acf_amd.detector import HipDetector
    H, W, B = args.height, args.width, args.batch
    model = synth.make_model(seed=1, name=args.model)
    base = torch.from_numpy(np.stack([synth.make_frame(i + 1, H, W, "luv") for i in range(4)])).cuda()
    frames = torch.empty((B, 3, W, H), dtype=torch.float32, device="cuda")
    for i in range(B):
        frames[i] = torch.roll(base[i % 4], shifts=(37 * (i // 4), 53 * (i // 4)), dims=(1, 2))
    det = HipDetector(model, H, W, 3, max_batch=B, max_hits=8192)
    for kv in args.opt:
        k, v = kv.split("=")
        det.set_option(k, int(v))
    det.set_option("profile", 1)
    det.pyramid(frames, B)
    det.detect()
    det.synchronize()
    det.profile()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        det.detect()
    det.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    prof = det.profile()
    print("== %s batch %d  detect() wall %.3f ms = %.2f us/frame" % (args.tag, B, dt * 1e3, dt * 1e6 / B))
    for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("   %-28s %8.3f ms/launch  %7.2f us/frame  (%d launches)" % (k, ms / max(n, 1), ms / max(n, 1) * 1e3 / B, n))
    if args.full:
        det.profile()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            det.run(frames, B)
        det.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        prof = det.profile()
        print("== %s run() wall %.3f ms = %.2f us/frame = %.0f FPS" % (args.tag, dt * 1e3, dt * 1e6 / B, B / dt))
        for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            print("   %-28s %8.3f ms/launch  %7.2f us/frame  (%d launches)" % (k, ms / max(n, 1), ms / max(n, 1) * 1e3 / B, n))
    hits = []
    for f in range(min(B, 8)):
        d, _ = det.detections(f)
        order = np.lexsort((d["y"], d["x"], d["scale"]))
        hits.append(np.stack([d["x"][order], d["y"][order], d["w"][order], d["h"][order], d["scale"][order],
                              d["score"][order].view(np.int32)]).astype(np.int64))
    if args.save:
        np.savez(args.save, **{"f%d" % i: h for i, h in enumerate(hits)})
    if args.check:
        ref = np.load(args.check)
        ok = all(np.array_equal(ref["f%d" % i], h) for i, h in enumerate(hits))
        print("   parity vs %s: %s (%s detections in frame 0)" % (os.path.basename(args.check), "OK" if ok else "MISMATCH", hits[0].shape[1]))
        if not ok:
            sys.exit(3)


if __name__ == "__main__":
    main()
