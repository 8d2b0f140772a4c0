"""BASELINE.json's configurations at FULL size, one frame each, bit-exact against the oracle (the oracle needs
0.1-2 s per frame at these sizes): every level of the fused pyramid, hit positions, score bits and mapped boxes.

 cfg 1  640x480 gray, FACE64 (7 channels, colour disabled), 2048 trees
 cfg 2  1920x1080 planar LUV, FACE80, 2048 trees (the benchmark workload)
 cfg 3  a batch of cfg-2 frames (8 here): per-frame results independent of batch position
 cfg 4  640x480 RGB, INRIA 128x64 model, pad [16 12], nOctUp 1 (RGB->LUV on the device, up-sampled real scale)
 cfg 5  3840x2160 planar LUV, 12 scales/octave, nApprox 11, FACE80 — plain, and with its LDCF post-stage (k = 4 5x5 filters per
        channel; no reference counterpart, checked against the oracle's restatement of the toolbox definition)
"""
import numpy as np
import pytest

from acf_amd import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


CFG = {
    "cfg1_vga_gray_face64": (480, 640, "gray", 1, dict(name="FACE64")),
    "cfg2_1080p_luv_face80": (1080, 1920, "luv", 3, dict(name="FACE80")),
    "cfg4_vga_rgb_inria": (480, 640, "rgb", 3, dict(name="INRIA")),
    "cfg5_4k_luv_face80_12perOct": (2160, 3840, "luv", 3, dict(name="FACE80", nPerOct=12, nApprox=11)),
}


@pytest.mark.parametrize("cfg", list(CFG))
def test_full_size_config_bit_exact(oracle, cfg):
    import torch
    from acf_amd.detector import HipDetector
    H, W, kind, d_in, kw = CFG[cfg]
    model = synth.make_model(seed=1, **kw)
    frame = synth.make_frame(2, H, W, kind)
    det = HipDetector(model, H, W, d_in, max_batch=1, max_hits=1 << 16)
    plan = oracle.Plan(model, H, W, d_in)
    assert len(det.levels) == plan.nScales
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    want, wh = oracle.detect(plan, pyr)
    det.run(torch.from_numpy(frame[None]).cuda())
    for i in range(plan.nScales):
        l = plan.levels[i]
        n = plan.nChns * l.hP * l.wP
        assert np.array_equal(bits(det.read_level(0, i)).ravel(), bits(pyr[l.offset:l.offset + n])), (cfg, "level", i)
    got, gh = det.detections(0)
    assert len(want) > 0, "the synthetic model should fire on this frame"
    assert gh.tobytes() == wh.tobytes()
    assert got.tobytes() == want.tobytes()
    det.close()


# fused_grad 2: gradMag inside the smoothing chain at every scale (what batches >= 16 frames get at scale 0); fused_tri 2: convTri's x pass
# on that chain as well (k_smooth_grad_tri: what batches >= 64 frames get)
@pytest.mark.parametrize("fused_grad,fused_tri", [(1, 0), (2, 0), (2, 2)])
def test_cfg3_batch_of_1080p_frames(oracle, fused_grad, fused_tri):
    import torch
    from acf_amd.detector import HipDetector
    H, W = 1080, 1920
    model = synth.make_model(seed=1, name="FACE80")
    n = 8
    frames = np.stack([synth.make_frame(100 + i, H, W, "luv") for i in range(n)])
    det = HipDetector(model, H, W, 3, max_batch=n, max_hits=1 << 15)
    det.set_option("fused_grad", fused_grad)
    det.set_option("fused_tri", fused_tri)
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, 3)
    for f in (0, 3, 7):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, wh = oracle.detect(plan, pyr)
        got, gh = det.detections(f)
        assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes(), f
    det.close()


def test_cfg5_ldcf_full_size(oracle):
    """cfg 5 as BASELINE.json names it: 4K, 12 scales/octave, FACE80 + k 5x5 filters per channel (LDCF).  58 filtered, halved levels
    of 40 channels and the cascade over them, against oracle/acf_oracle.c's acfo_ldcf_* (the only definition there is)."""
    import torch
    from acf_amd import capi
    from acf_amd.detector import HipDetector
    H, W = 2160, 3840
    model = synth.make_model(seed=1, name="FACE80", nPerOct=12, nApprox=11, ldcfK=4, cascThr=-2.0, nTrees=512)
    frame = synth.make_frame(2, H, W, "luv")
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 17)
    plan = oracle.Plan(model, H, W, 3)
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    lvL, pyrL, k = oracle.ldcf(plan, pyr)
    want, wh = oracle.detect_ldcf(plan, lvL, pyrL, cap=1 << 17)
    det.run(torch.from_numpy(frame[None]).cuda())
    for i in range(plan.nScales):
        l = lvL[i]
        n = plan.nChns * k * l.hP * l.wP
        got = det.read_tap(0, capi.TAP_LDCF, i, (plan.nChns * k, l.wP, l.hP))
        assert np.array_equal(bits(got).ravel(), bits(pyrL[l.offset:l.offset + n])), ("LDCF level", i)
    got, gh = det.detections(0)
    assert len(want) > 0
    assert gh.tobytes() == wh.tobytes()
    assert got.tobytes() == want.tobytes()
    det.close()


TREF_CFG = {
    "cfg1_vga_gray_face64": ("gray", "FACE64"),
    "cfg2_1080p_luv_face80": ("luv", "FACE80"),
    "cfg4_vga_rgb_inria": ("rgb", "INRIA"),
}


@pytest.mark.parametrize("cfg", list(TREF_CFG))
def test_hip_hits_are_the_texact_side_of_the_tref_study(cfg):
    """tests/golden/tref_study.npz holds, per frame, the cascade's hits of the oracle (T-exact) and of the reference's own compiled
    SSE kernels under the same orchestration (T-ref; made in the build container by tests/golden/make_tref.py).  The HIP path must
    be the T-exact side bit for bit — then DESIGN.md section 2's T-ref table (windows in one tier only, score differences) is a
    statement about the HIP path against the reference's arithmetic, measured rather than argued."""
    import os
    import torch
    from acf_amd.detector import HipDetector
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tref_study.npz"))
    H, W, d_in, nframes, seed0, mseed = [int(v) for v in fix[cfg + "_meta"]]
    kind, preset = TREF_CFG[cfg]
    model = synth.make_model(seed=mseed, name=preset)
    n = min(nframes, 4)
    frames = np.stack([synth.make_frame(seed0 + f, H, W, kind) for f in range(n)])
    det = HipDetector(model, H, W, d_in, max_batch=n, max_hits=1 << 15)
    det.run(torch.from_numpy(frames).cuda())
    for f in range(n):
        _, gh = det.detections(f)
        want = fix["%s_f%d_hits_exact" % (cfg, f)]
        assert gh.tobytes() == want.tobytes(), (cfg, f)
        ref = fix["%s_f%d_hits_ref" % (cfg, f)]
        kg = (gh["scale"].astype(np.int64) << 40) | (gh["c"].astype(np.int64) << 20) | gh["r"].astype(np.int64)
        kr = (ref["scale"].astype(np.int64) << 40) | (ref["c"].astype(np.int64) << 20) | ref["r"].astype(np.int64)
        common = np.intersect1d(kg, kr)
        assert len(common) >= 0.95 * max(len(kg), len(kr)), (cfg, f, len(kg), len(kr), len(common))
    det.close()
