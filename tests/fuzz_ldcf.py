"""Random geometries of the LDCF post-stage (k_ldcf_tile: k 5x5 filters per channel, two at a time, + the halving) against the
oracle's restatement (oracle/acf_oracle.c acfo_ldcf_*): filtered + halved levels bit for bit, hits, boxes and scores.

    python tests/fuzz_ldcf.py [seed] [cases]

Frame sizes that give odd and even level sizes (the halving's generic round(.5 n) geometry and its exact form), k = 1 .. 5 (odd k: the
last pair's second filter is a dummy), strides 4 and 8, several scales per octave."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from acf_amd import capi, synth
from acf_amd.detector import HipDetector
from oracle import binding as ob

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


bad = ran = checked = 0
for it in range(N):
    H = int(rng.randint(24, 150)) * 4 if rng.rand() < 0.6 else int(rng.randint(100, 620))
    W = int(rng.randint(24, 200)) * 4 if rng.rand() < 0.6 else int(rng.randint(100, 800))
    k = int(rng.choice([1, 2, 2, 3, 4, 4, 5]))
    kw = dict(name="TINY", nTrees=int(rng.choice([32, 64, 128])), ldcfK=k, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32,
              minDs_h=32, minDs_w=32, cascThr=-2.0, nPerOct=int(rng.choice([4, 8, 12])), stride=int(rng.choice([4, 4, 8])))
    kw["nApprox"] = int(rng.choice([0, kw["nPerOct"] - 1]))
    nF = int(rng.choice([1, 2]))
    try:
        model = synth.make_model(seed=int(rng.randint(1, 99)), **kw)
        frames = np.stack([synth.make_frame(int(rng.randint(1, 9999)), H, W, "luv") for _ in range(nF)])
        plan = ob.Plan(model, H, W, 3)
        det = HipDetector(model, H, W, 3, max_batch=nF, max_hits=1 << 16)
    except Exception as e:  # geometry the plan refuses (too small for the model): not a parity case
        continue
    det.run(torch.from_numpy(frames).cuda())
    ok = True
    for f in range(nF):
        pyr, _, _ = ob.chns_pyramid(plan, frames[f])
        lvL, pyrL, kk = ob.ldcf(plan, pyr)
        for i in range(plan.nScales):
            l = lvL[i]
            n = plan.nChns * kk * l.hP * l.wP
            want = pyrL[l.offset:l.offset + n].reshape(plan.nChns * kk, l.wP, l.hP)
            got = det.read_tap(f, capi.TAP_LDCF, i, (plan.nChns * kk, l.wP, l.hP))
            if not np.array_equal(bits(got), bits(want)):
                ok = False
                print("LDCF level differs", i, l.hP, l.wP, float(np.abs(got - want).max()))
                break
        want, wh = ob.detect_ldcf(plan, lvL, pyrL)
        if len(want) >= (1 << 16):
            continue
        got, gh = det.detections(f)
        checked += 1
        if got.tobytes() != want.tobytes() or gh.tobytes() != wh.tobytes():
            ok = False
    det.close()
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", H, W, nF, kw)
print("cases", N, "ran", ran, "frames_checked", checked, "mismatches", bad)
sys.exit(1 if bad else 0)
