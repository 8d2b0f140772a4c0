#!/usr/bin/env python
"""Regenerate the golden vectors under tests/golden/.

Run in the build container (needs /root/reference for oracle/_ref):

    python tests/golden/make_golden.py

Two kinds of fixture, both plain data (inputs + expected outputs):

 ref_ops.npz       outputs of the REFERENCE's own toolbox kernels — compiled
                   unmodified from /root/reference/src/lib/acf/acf/toolbox/
                   {convConst,gradientMex,wrappers}.cpp into
                   oracle/_ref/libacfref.so (oracle/Makefile) — on seeded
                   inputs.  These pin the oracle (and, on the GPU box where
                   /root/reference and possibly _ref are absent, the HIP
                   kernels) to the reference's real bits:
                     convTri1 (separate src/dst and the pyramid's in-place
                     aliased call), convTri r=5 / r=2, grad2, gradHist
                     -> bit-exact targets;
                     gradMag M/O and gradMagNorm -> targets within the
                     _mm_rsqrt_ps/_mm_rcp_ps bound.
 ref_resample_luv.npz  the same for `resample` (imResampleMex.cpp) and rgbConvert's
                   rgb2luv / rgb2luv_sse / rgb2gray (rgbConvertMex.cpp), whose
                   OpenCV-free bodies the Makefile takes from the reference
                   files by line range (oracle/ref_api_resample.cpp,
                   oracle/ref_api_rgbconvert.cpp state the ranges).
 pipeline_*.npz    end-to-end vectors of the restated orchestration (oracle/
                   acf_oracle.c): frame, model arrays, scales, every level of
                   the fused pyramid, cascade hits and mapped boxes.  The
                   reference's tests hold no vectors for this part (SURVEY.md
                   §4), so these freeze the restatement that was validated
                   stage by stage against _ref at generation time (the script
                   asserts that before writing).
 scales.npz        Detector::getScales for the five BASELINE.json config sizes.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from acf_amd import capi, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

F = capi.fptr


def rnd(seed, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * synth.uniform(seed, n, 77)).astype(np.float32).reshape(shape)


def ref_ops():
    assert ob.have_ref(), "oracle/_ref/libacfref.so missing: run `make -C oracle` with /root/reference present"
    r = ob.ref()
    o = ob.lib()
    out = {}
    cases = [(48, 36), (64, 52), (37, 29)]  # (h, w); the last is odd-sized: the SSE tails / unaligned paths
    out["sizes"] = np.asarray(cases, dtype=np.int32)
    for k, (h, w) in enumerate(cases):
        a = ob.aligned_copy(rnd(100 + k, (3, w, h)))
        out["in%d" % k] = np.array(a)
        # convTri1, p = 2 (r = 1), separate destination
        b = ob.aligned((3, w, h))
        r.ref_convTri1(F(a), F(b), h, w, 3, 2.0, 1)
        out["tri1_%d" % k] = np.array(b)
        # the pyramid's in-place call (chnsCompute.cpp:239): source == destination
        c = ob.aligned_copy(a)
        r.ref_convTri1(F(c), F(c), h, w, 3, 2.0, 1)
        out["tri1_aliased_%d" % k] = np.array(c)
        for rad in (5, 2):
            if w <= 2 * rad or h <= 2 * rad:
                continue
            b = ob.aligned((1, w, h))
            a1 = ob.aligned_copy(a[:1])
            r.ref_convTri(F(a1), F(b), h, w, 1, rad, 1)
            out["tri_r%d_%d" % (rad, k)] = np.array(b[0])
        a1 = ob.aligned_copy(a[0])
        gx, gy = ob.aligned((w, h)), ob.aligned((w, h))
        r.ref_grad2(F(a1), F(gx), F(gy), h, w, 1)
        out["gx_%d" % k], out["gy_%d" % k] = np.array(gx), np.array(gy)
        M, O = ob.aligned((w, h)), ob.aligned((w, h))
        r.ref_gradMag(F(a1), F(M), F(O), h, w, 1, 0)
        out["M_%d" % k], out["O_%d" % k] = np.array(M), np.array(O)
        if h % 4 == 0 and w % 4 == 0:
            # gradHist on the reference's own M, O: bin 4, 6 orientations, softBin 0
            H = ob.aligned((6, w // 4, h // 4))
            H[...] = 0
            r.ref_gradHist(F(M), F(O), F(H), h, w, 4, 6, 0, 0)
            out["H_%d" % k] = np.array(H)
        S = ob.aligned((w, h))
        if w > 10 and h > 10:
            r.ref_convTri(F(M), F(S), h, w, 1, 5, 1)
            Mn = ob.aligned_copy(M)
            r.ref_gradMagNorm(F(Mn), F(S), h, w, np.float32(0.005))
            out["S_%d" % k], out["Mn_%d" % k] = np.array(S), np.array(Mn)
        # the oracle must agree with these before they are frozen
        t = np.zeros((3, w, h), np.float32)
        o.acfo_conv_tri1(F(np.ascontiguousarray(a)), F(t), h, w, 3, np.float32(2.0), 1)
        assert np.array_equal(t.view(np.uint32), out["tri1_%d" % k].view(np.uint32))
        t = np.ascontiguousarray(a).copy()
        o.acfo_conv_tri1(F(t), F(t), h, w, 3, np.float32(2.0), 1)
        assert np.array_equal(t.view(np.uint32), out["tri1_aliased_%d" % k].view(np.uint32))
    np.savez_compressed(os.path.join(HERE, "ref_ops.npz"), **out)
    return out


# geometries (ha, wa, hb, wb, d, gain): exact /2 /3 /4, generic down (2..4 taps and the >4-tap scatter form), up-sampling, mixed
# axes, odd sizes, and the shapes the pyramid uses (an approximated channel level with its power-law gain)
RESAMPLE_CASES = [(64, 48, 32, 24, 3, 1.0), (66, 48, 22, 16, 1, 1.0), (64, 48, 16, 12, 1, 1.0), (64, 48, 57, 43, 3, 1.0), (63, 50, 31, 27, 3, 1.0),
                  (120, 160, 13, 17, 1, 1.0), (32, 24, 64, 48, 3, 1.0), (31, 27, 63, 50, 1, 1.0), (64, 48, 32, 60, 1, 1.0), (37, 41, 37, 41, 1, 1.0),
                  (68, 120, 62, 110, 4, 0.8705506), (68, 120, 74, 132, 4, 1.3195079), (136, 240, 68, 121, 3, 1.0)]


def ref_resample_luv():
    """ref_resample_luv.npz: outputs of the reference's own `resample` (imResampleMex.cpp:122-383) and rgbConvert /
    rgb2luv / rgb2luv_sse / rgb2gray bodies (rgbConvertMex.cpp:17-380), spliced by line range into oracle/_ref/libacfref.so
    (oracle/Makefile), on seeded inputs.  resample, rgb2gray and the scalar rgb2luv are bit-exact targets; the vector
    rgb2luv (n % 4 == 0, one _mm_rcp_ps) is a target within the rcp bound with L exact."""
    assert ob.have_ref(), "oracle/_ref/libacfref.so missing: run `make -C oracle` with /root/reference present"
    r = ob.ref()
    o = ob.lib()
    out = {"resample_cases": np.asarray(RESAMPLE_CASES, dtype=np.float64)}
    for k, (ha, wa, hb, wb, d, g) in enumerate(RESAMPLE_CASES):
        a = ob.aligned_copy(rnd(500 + k, (d, wa, ha)))
        b, t = ob.aligned((d, wb, hb)), ob.aligned((d, wb, hb))
        r.ref_resample(F(a), F(b), ha, hb, wa, wb, d, np.float32(g))
        assert o.acfo_resample(F(a), F(t), ha, hb, wa, wb, d, np.float32(g)) == 0
        assert np.array_equal(b.view(np.uint32), t.view(np.uint32))  # the oracle agrees before anything is frozen
        out["rs_out%d" % k] = np.array(b)  # input: rnd(500 + k, (d, wa, ha)) (seeded: not stored)
    luv = [(48, 36), (37, 29), (64, 51)]  # (h, w): n % 4 == 0 -> vector body; the other two -> scalar body
    out["luv_sizes"] = np.asarray(luv, dtype=np.int32)
    for k, (h, w) in enumerate(luv):
        a = ob.aligned_copy(synth.make_frame(600 + k, h, w, "rgb"))
        b, t, gr, tg = ob.aligned((3, w, h)), ob.aligned((3, w, h)), ob.aligned((w, h)), ob.aligned((w, h))
        assert r.ref_rgbConvert(F(a), F(b), h * w, 3, 2, np.float32(1.0)) == 0
        assert r.ref_rgbConvert(F(a), F(gr), h * w, 3, 0, np.float32(1.0)) == 0
        o.acfo_rgb2luv(F(a), F(t), h * w)
        o.acfo_rgb2gray(F(a), F(tg), h * w)
        assert np.array_equal(gr.view(np.uint32), tg.view(np.uint32)) and np.array_equal(b[0].view(np.uint32), t[0].view(np.uint32))
        if (h * w) % 4:
            assert np.array_equal(b.view(np.uint32), t.view(np.uint32))
        out["luv_out%d" % k], out["gray_out%d" % k] = np.array(b), np.array(gr)  # input: synth.make_frame(600 + k, h, w, "rgb")
    np.savez_compressed(os.path.join(HERE, "ref_resample_luv.npz"), **out)
    return out


PIPELINES = {
    # name: (H, W, kind, d_in, frame seed, model kwargs)
    "tiny_luv": (64, 80, "luv", 3, 17, dict(name="TINY", nTrees=96, seed=3)),
    "rgb_inria": (112, 96, "rgb", 3, 21, dict(name="INRIA", nTrees=64, seed=5, cascThr=-0.6)),
    "gray_face64": (96, 128, "gray", 1, 23, dict(name="FACE64", nTrees=64, seed=7, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32, minDs_h=32, minDs_w=32)),
    "depth0": (72, 96, "luv", 3, 29, dict(name="TINY", nTrees=80, seed=9, treeDepth=0)),
    # LDCF post-stage (BASELINE cfg 5; no reference counterpart): the frozen oracle output doubles as a regression vector
    "ldcf_k3": (96, 128, "luv", 3, 31, dict(name="TINY", nTrees=64, seed=3, ldcfK=3, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32,
                                           minDs_h=32, minDs_w=32, cascThr=-2.0)),
}


def pipelines():
    for name, (H, W, kind, d_in, fseed, kw) in PIPELINES.items():
        kw = dict(kw)
        seed = kw.pop("seed")
        model = synth.make_model(seed=seed, **kw)
        frame = synth.make_frame(fseed, H, W, kind)
        plan = ob.Plan(model, H, W, d_in)
        pyr, _, _ = ob.chns_pyramid(plan, frame)
        extra = {}
        if int(model.get("ldcfK", 0)) > 0:
            lvL, pyrL, _ = ob.ldcf(plan, pyr)
            det, hits = ob.detect_ldcf(plan, lvL, pyrL)
            extra = dict(ldcf_filters=model["ldcfFilters"], ldcf_pyramid=pyrL,
                         ldcf_geom=np.asarray([(lvL[i].hP, lvL[i].wP, lvL[i].nWinR, lvL[i].nWinC, lvL[i].offset) for i in range(plan.nScales)], dtype=np.int64))
        else:
            det, hits = ob.detect(plan, pyr)
        lv = plan.levels
        out = dict(
            H=np.int32(H), W=np.int32(W), d_in=np.int32(d_in), frame=frame,
            fids=model["fids"], thrs=model["thrs"], hs=model["hs"], child=model["child"],
            opts_json=np.asarray(json.dumps({k: v for k, v in sorted(model.items()) if k not in ("fids", "thrs", "hs", "child", "ldcfFilters")})),
            scales=np.asarray([lv[i].scale for i in range(plan.nScales)]),
            scaleshw=np.asarray([(lv[i].scalehw_h, lv[i].scalehw_w) for i in range(plan.nScales)]),
            level_geom=np.asarray([(lv[i].isReal, lv[i].realIndex, lv[i].hC, lv[i].wC, lv[i].hP, lv[i].wP, lv[i].nWinR, lv[i].nWinC, lv[i].offset)
                                   for i in range(plan.nScales)], dtype=np.int64),
            pyramid=pyr, det=det, hits=hits, **extra)
        assert len(det) > 0, name
        np.savez_compressed(os.path.join(HERE, "pipeline_%s.npz" % name), **out)
        print(name, "levels", plan.nScales, "pyramid floats", pyr.size, "detections", len(det))


def scales():
    cfgs = [  # (H, W, nPerOct, nOctUp, minDs_h, minDs_w, shrink) — BASELINE.json configs 1..5
        (480, 640, 8, 0, 64, 64, 4), (1080, 1920, 8, 0, 80, 80, 4), (1080, 1920, 8, 0, 80, 80, 4),
        (480, 640, 8, 1, 100, 41, 4), (2160, 3840, 12, 0, 80, 80, 4)]
    out = {"cfgs": np.asarray(cfgs, dtype=np.int32)}
    o = ob.lib()
    import ctypes as C
    for k, (H, W, npo, nou, mh, mw, sh) in enumerate(cfgs):
        s = (C.c_double * 256)()
        a = (C.c_double * 256)()
        b = (C.c_double * 256)()
        n = o.acfo_get_scales(npo, nou, mh, mw, sh, H, W, s, a, b, 256)
        out["scales_%d" % k] = np.asarray(s[:n])
        out["shw_h_%d" % k] = np.asarray(a[:n])
        out["shw_w_%d" % k] = np.asarray(b[:n])
    np.savez_compressed(os.path.join(HERE, "scales.npz"), **out)


if __name__ == "__main__":
    ob.build()
    if "--only-new" not in sys.argv:  # (the older fixtures are deterministic; this only saves time)
        ref_ops()
    ref_resample_luv()
    if "--only-new" in sys.argv:
        sys.exit(0)
    pipelines()
    scales()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
