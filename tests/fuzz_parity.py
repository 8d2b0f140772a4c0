"""Random geometries / options against the oracle (pyramid bits, hits, boxes): a one-off sweep beside the fixed cases of tests/."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from acf_amd import synth
from acf_amd.detector import HipDetector
from oracle import binding as ob
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
ran = 0
checked = 0  # frames whose pyramid, hits and boxes were actually compared
for it in range(N):
    H = int(rng.randint(20, 180)) * 4 if rng.rand() < 0.8 else int(rng.randint(80, 600))
    W = int(rng.randint(20, 200)) * 4 if rng.rand() < 0.8 else int(rng.randint(80, 700))
    depth = int(rng.choice([1, 2, 2, 3, 4]))
    nTrees = int(rng.choice([20, 64, 96, 160, 300]))
    kw = dict(name="TINY", nTrees=nTrees, treeDepth=depth, cascThr=-1.0 if nTrees < 300 else -3.0,
              nPerOct=int(rng.choice([4, 8, 8, 12])), nApprox=int(rng.choice([0, 3, 7, -1])), full=int(rng.rand() < 0.2),
              colorChn=int(rng.choice([0, 0, 1, 2])), pad_h=int(rng.choice([0, 0, 4, 8])), pad_w=int(rng.choice([0, 0, 4, 12])),
              softBin=int(rng.choice([0, 0, 0, -2, 2])), stride=int(rng.choice([4, 4, 4, 4, 8, 2, 1])))  # (stride < shrink: one evaluation per distinct offset)
    if kw["nApprox"] < 0:
        kw["nApprox"] = kw["nPerOct"] - 1
    kind = "rgb" if rng.rand() < 0.3 else "luv"   # RGB frames go through rgbConvert (k_rgb2luv) first
    kw["isLuv"] = int(kind == "luv")
    arith = int(rng.rand() < 0.25)                # the reference's rcpps / rsqrtps bits (option arith; oracle: acfo_set_approx(3))
    nF = int(rng.choice([1, 2, 3]))
    opts = dict(fused_grad=int(rng.choice([0, 1, 2, 2])), fused_tri=int(rng.choice([0, 1, 2, 2])), smooth_segments=int(rng.choice([0, 1, 3, 5])), smooth_warm=int(rng.choice([16, 32, 96])),
                scale_streams=int(rng.rand() < 0.5), keep_pyramid=int(rng.rand() < 0.7), rank_cells=int(rng.rand() < 0.7), graph=int(rng.rand() < 0.3),
                cascade_tiles=int(rng.rand() < 0.8), level_segments=int(rng.choice([0, 1, 4])), tile_persist=int(rng.choice([0, 1, 1, 8])))
    try:
        model = synth.make_model(seed=int(rng.randint(1, 99)), **kw)
        frames = np.stack([synth.make_frame(int(rng.randint(1, 9999)), H, W, kind) for _ in range(nF)])
        if rng.rand() < 0.3:
            frames[0, :, : W // 3, H // 4: H // 2] = 0.0
        plan = ob.Plan(model, H, W, 3)
    except Exception as e:  # geometry the plan refuses (too small for the model): not a parity case
        continue
    try:
        det = HipDetector(model, H, W, 3, max_batch=nF, max_hits=1 << 15)
    except Exception as e:
        print("plan refused", H, W, kw, str(e)[:80])
        continue
    for k, v in opts.items():
        det.set_option(k, v)
    if arith:
        det.set_x86_tables(*ob.x86_fixture())
        det.set_option("arith", 1)
        ob.set_x86_tables(*ob.x86_fixture())
    dev = torch.from_numpy(frames).cuda()
    ok = True
    for rep in range(2 if opts["graph"] else 1):
        det.run(dev, nF)
        for f in range(nF):
            ob.set_approx(3 if arith else 0)
            try:
                pyr, _, _ = ob.chns_pyramid(plan, frames[f])
            finally:
                ob.set_approx(0)
            try:
                want, wh = ob.detect(plan, pyr)
            except RuntimeError:
                continue  # (more hits than the oracle binding's buffer)
            if len(want) >= (1 << 15):
                continue  # (more hits than the plan's capacity: the library reports that as an error)
            got, gh = det.detections(f)
            checked += 1
            # (keep_pyramid = 0: the float levels may not exist — detection-only call; hits, boxes and scores still must)
            pyr_ok = (not opts["keep_pyramid"]) or np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32))
            if not pyr_ok or got.tobytes() != want.tobytes() or gh.tobytes() != wh.tobytes():
                ok = False
    det.close()
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", H, W, nF, kind, "arith", arith, kw, opts)
print("cases", N, "ran", ran, "frames_checked", checked, "mismatches", bad)
