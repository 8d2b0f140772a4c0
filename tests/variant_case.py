"""One process per setting (the environment is read once per process; options come in VARIANT_OPTIONS): a VGA frame pair through
the library against the oracle — every pyramid level's bits, hits, boxes and scores — for a depth-2 model with a tail
(300 trees, low cascThr: windows reach the leaf-code stages), and an LDCF model when `ldcf` is given.  tests/test_gpu_variants.py
runs this file under each setting; exit status 0 = identical."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acf_amd import synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402
from oracle import binding as ob  # noqa: E402


def main():
    ldcf = len(sys.argv) > 1 and sys.argv[1] == "ldcf"
    H, W, nF = 480, 640, 2
    kw = dict(name="FACE80", nTrees=300, cascThr=-3.0)
    if len(sys.argv) > 1 and sys.argv[1].startswith("depth"):
        d = int(sys.argv[1][5:])
        kw = dict(name="FACE80", nTrees=300, cascThr=-3.0 if d == 1 else -2.0, treeDepth=d)
    if ldcf:
        kw = dict(name="FACE80", nTrees=256, ldcfK=4, cascThr=-2.0)  # (tests/test_gpu_pipeline.py, face80_k4_vga)
    model = synth.make_model(seed=11, **kw)
    frames = np.stack([synth.make_frame(900 + i, H, W, "luv") for i in range(nF)])
    plan = ob.Plan(model, H, W, 3)
    det = HipDetector()
    for kv in filter(None, os.environ.get("VARIANT_OPTIONS", "").split(",")):   # options of the setting, before the plan
        k, v = kv.split("=")
        det.set_option(k, int(v))
    det.set_model(model)
    det.plan(H, W, 3, max_batch=nF, max_hits=1 << 17)
    det.run(torch.from_numpy(frames).cuda(), nF)
    total = 0
    for f in range(nF):
        pyr, _, _ = ob.chns_pyramid(plan, frames[f])
        if ldcf:
            lvL, pyrL, _ = ob.ldcf(plan, pyr)
            want, wh = ob.detect_ldcf(plan, lvL, pyrL)
        else:
            want, wh = ob.detect(plan, pyr, cap=1 << 17)
        got, gh = det.detections(f)
        if not np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)):
            print("pyramid differs, frame", f)
            return 1
        if got.tobytes() != want.tobytes() or gh.tobytes() != wh.tobytes():
            print("detections differ, frame", f, len(got), len(want))
            return 1
        total += len(want)
    det.close()
    if total == 0:
        print("no detections: the case checks nothing")
        return 1
    print("ok", total)
    return 0


if __name__ == "__main__":
    sys.exit(main())
