"""Image-specific lambdas (chnsPyramid.cpp:341-374): a model without lambdas gets them from every
image — the mean of each channel type at two real scales, lambda = -log2(f0/f1) / log2(s0/s1).

Tolerance, stated once: the reference's sum(MatP) is cv::sum (f32 data, f64 accumulation) in OpenCV's own SIMD order,
which cannot be restated without OpenCV; the oracle and the device share ONE order of the f64 additions (256 interleaved
partial sums + a binary tree: oracle/acf_oracle.c:acfo_plane_sum, kernels.hip.h:k_plane_sums), so the device is compared
with the oracle bit for bit, and the oracle with a plain numpy f64 sum (yet another order) to 1e-12 relative — the
bound that covers any order of ~4e5 f64 additions of f32 values (each reorder moves the sum by a few ulp, 1e-16)."""
import ctypes as C

import numpy as np
import pytest

from acf_amd import capi, synth

CFG = dict(name="TINY", nTrees=128, lambdas=[], cascThr=-2.0)
E_INVALID = 1  # ACF_HIP_E_INVALID (include/acf_hip.h)
H, W = 240, 320


def _numpy_lambdas(plan, chns, model):
    i0, i1 = C.c_int(), C.c_int()
    from oracle import binding as ob
    assert ob.lib().acfo_lambda_levels(C.byref(plan.params), plan.nScales, C.byref(i0), C.byref(i1)) == 1
    lv = plan.levels
    d = 1 if model["colorSpace"] == capi.CS_GRAY else 3
    n = [d if model["colorEnabled"] else 0, 1 if model["gradMagEnabled"] else 0, model["nOrients"] if model["gradHistEnabled"] else 0]
    out, z = [], 0
    for k in n:
        if k == 0:
            out.append(0.0)
            continue
        a, b = chns[i0.value][z:z + k].astype(np.float64), chns[i1.value][z:z + k].astype(np.float64)
        f0, f1 = a.sum() / a.size, b.sum() / b.size
        out.append(float(-(np.log(f0 / f1) / np.log(2.0)) / (np.log(lv[i0.value].scale / lv[i1.value].scale) / np.log(2.0))))
        z += k
    return out, (i0.value, i1.value)


def test_oracle_lambdas_match_the_formula(oracle):
    model = synth.make_model(seed=3, **CFG)
    plan = oracle.Plan(model, H, W, 3)
    frame = synth.make_frame(5, H, W, "luv")
    _, _, chns = oracle.chns_pyramid(plan, frame, want_chns=True)
    got = oracle.last_lambdas()
    want, (i0, i1) = _numpy_lambdas(plan, chns, model)
    # 28 scales, real levels 0, 8, 16, 24: more than two candidates -> the second and third (chnsPyramid.cpp:352-355)
    assert (i0, i1) == (8, 16)
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-12 * max(1.0, abs(w)), (g, w)
    # the blocked plane sum against numpy's pairwise f64 sum
    x = chns[8][3]
    s = oracle.lib().acfo_plane_sum(x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    assert abs(s - x.astype(np.float64).sum()) <= 1e-12 * abs(s)


def test_lambdas_change_the_approximated_levels_only(oracle):
    """Real levels do not depend on the lambdas; approximated ones follow pow(scale ratio, -lambda) (chnsPyramid.cpp:393)."""
    model0 = synth.make_model(seed=3, **CFG)
    frame = synth.make_frame(5, H, W, "luv")
    plan0 = oracle.Plan(model0, H, W, 3)
    pyr0, _, _ = oracle.chns_pyramid(plan0, frame)
    lam = oracle.last_lambdas()
    model1 = synth.make_model(seed=3, **dict(CFG, lambdas=lam))
    plan1 = oracle.Plan(model1, H, W, 3)
    pyr1, _, _ = oracle.chns_pyramid(plan1, frame)
    # supplying the estimated lambdas explicitly reproduces the pyramid bit for bit
    assert np.array_equal(pyr0.view(np.uint32), pyr1.view(np.uint32))
    model2 = synth.make_model(seed=3, **dict(CFG, lambdas=[0.0, 0.1105, 0.1083]))
    pyr2, _, _ = oracle.chns_pyramid(oracle.Plan(model2, H, W, 3), frame)
    for i in range(plan0.nScales):
        same = np.array_equal(plan0.level_view(pyr0, i), plan0.level_view(pyr2, i))
        assert same == bool(plan0.levels[i].isReal), i


def test_plan_accepts_no_lambdas_and_rejects_too_few_scales():
    lib = capi.load()
    model = synth.make_model(seed=3, **CFG)
    params, _keep = capi.make_params(model)
    lv = (capi.Level * 64)()
    n, nc = C.c_int(), C.c_int()
    assert lib.acf_hip_plan_levels(C.byref(params), H, W, 3, lv, 64, C.byref(n), C.byref(nc)) == 0 and n.value == 28
    # 48 x 48 pixels, minDs 16: 9 scales = real levels 0 and 8 only -> two candidates, still fine
    assert lib.acf_hip_plan_levels(C.byref(params), 48, 48, 3, lv, 64, C.byref(n), C.byref(nc)) == 0 and n.value == 9
    # 40 x 40: fewer than 9 scales -> one real level: CV_Assert(is.size() >= 2) (chnsPyramid.cpp:351)
    assert lib.acf_hip_plan_levels(C.byref(params), 40, 40, 3, lv, 64, C.byref(n), C.byref(nc)) == E_INVALID
    # one or two lambdas is an error, not "estimate the rest"
    bad = synth.make_model(seed=3, **dict(CFG, lambdas=[0.1]))
    pb, _k = capi.make_params(bad)
    assert lib.acf_hip_plan_levels(C.byref(pb), H, W, 3, lv, 64, C.byref(n), C.byref(nc)) == E_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["tiny_luv", "gray_face64", "inria_up"])
def test_gpu_image_specific_lambdas(oracle, cfg):
    import torch
    from acf_amd.detector import HipDetector
    kind, d_in, kw = {
        "tiny_luv": ("luv", 3, CFG),
        "gray_face64": ("gray", 1, dict(name="FACE64", nTrees=128, lambdas=[])),
        "inria_up": ("rgb", 3, dict(name="INRIA", nTrees=128, lambdas=[])),  # nOctUp 1: the candidate levels start one octave in
    }[cfg]
    model = synth.make_model(seed=3, **kw)
    frames = [synth.make_frame(60 + i, H, W, kind) for i in range(3)]
    det = HipDetector(model, H, W, d_in, max_batch=3, max_hits=1 << 16)
    det.run(torch.from_numpy(np.stack(frames)).cuda())
    plan = oracle.Plan(model, H, W, d_in)
    lams = []
    for f, frame in enumerate(frames):
        pyr, _, _ = oracle.chns_pyramid(plan, frame)
        lam = oracle.last_lambdas()
        lams.append(lam)
        assert det.lambdas(f) == lam, (f, det.lambdas(f), lam)          # same additions in the same order: identical doubles
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), f
        want, _ = oracle.detect(plan, pyr)
        got, _ = det.detections(f)
        assert len(got) == len(want)
        for k in ("x", "y", "w", "h", "scale"):
            assert np.array_equal(got[k], want[k]), (f, k)
        assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32)), f
    assert lams[0] != lams[1] != lams[2]  # per image, not per batch
    # a model WITH lambdas reports its own
    m2 = synth.make_model(seed=3, **dict(kw, lambdas=[0.0, 0.1105, 0.1083]))
    det2 = HipDetector(m2, H, W, d_in, max_batch=1, max_hits=1 << 16)
    det2.run(torch.from_numpy(frames[0][None]).cuda())
    assert det2.lambdas(0) == [0.0, 0.1105, 0.1083]
