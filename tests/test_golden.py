"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

ref_ops.npz holds outputs of the reference's own toolbox kernels (compiled
unmodified into oracle/_ref at generation time); pipeline_*.npz hold the
end-to-end vectors.  CPU tests pin the oracle to them; the GPU tests pin the
HIP path (through the C ABI) to the same bytes, so the GPU box — which has no
/root/reference — still checks against the reference's real outputs.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from acf_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F = capi.fptr
PIPES = ["tiny_luv", "rgb_inria", "gray_face64", "depth0", "ldcf_k3"]
# _mm_rsqrt_ps / _mm_rcp_ps: relative error <= 1.5 * 2^-12 each (Intel SDM); gradMag chains both
RCP = 1.5 * 2.0 ** -12


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ref_ops.npz"))


def load_pipeline(name):
    z = np.load(os.path.join(GOLD, "pipeline_%s.npz" % name))
    model = json.loads(str(z["opts_json"]))
    for k in ("fids", "thrs", "hs", "child"):
        model[k] = z[k]
    if "ldcf_filters" in z.files:
        model["ldcfFilters"] = z["ldcf_filters"]
    return z, model


# ------------------------------------------------------------------ CPU: oracle vs golden

def test_oracle_matches_reference_kernel_outputs(oracle, ops):
    o = oracle.lib()
    for k, (h, w) in enumerate(ops["sizes"]):
        h, w = int(h), int(w)
        a = np.ascontiguousarray(ops["in%d" % k])
        t = np.zeros_like(a)
        assert o.acfo_conv_tri1(F(a), F(t), h, w, 3, np.float32(2.0), 1) == 0
        assert np.array_equal(bits(t), bits(ops["tri1_%d" % k]))
        t = a.copy()
        o.acfo_conv_tri1(F(t), F(t), h, w, 3, np.float32(2.0), 1)  # in place: the pyramid's aliased call
        assert np.array_equal(bits(t), bits(ops["tri1_aliased_%d" % k]))
        for rad in (5, 2):
            key = "tri_r%d_%d" % (rad, k)
            if key in ops:
                t = np.zeros((w, h), np.float32)
                assert o.acfo_conv_tri(F(np.ascontiguousarray(a[0])), F(t), h, w, 1, rad, 1) == 0
                assert np.array_equal(bits(t), bits(ops[key])), key
        gx, gy = np.zeros((w, h), np.float32), np.zeros((w, h), np.float32)
        o.acfo_grad2(F(np.ascontiguousarray(a[0])), F(gx), F(gy), h, w, 1)
        assert np.array_equal(bits(gx), bits(ops["gx_%d" % k])) and np.array_equal(bits(gy), bits(ops["gy_%d" % k]))
        M, O = np.zeros((w, h), np.float32), np.zeros((w, h), np.float32)
        o.acfo_grad_mag(F(np.ascontiguousarray(a[0])), F(M), F(O), h, w, 1, 0)
        refM = ops["M_%d" % k]
        assert np.all(np.abs(M - refM) <= 2.2 * RCP * np.abs(refM) + 1e-12)
        if "H_%d" % k in ops:
            H = np.zeros((6, w // 4, h // 4), np.float32)
            assert o.acfo_grad_hist(F(np.ascontiguousarray(refM)), F(np.ascontiguousarray(ops["O_%d" % k])), F(H), h, w, 4, 6, 0, 0) == 0
            assert np.array_equal(bits(H), bits(ops["H_%d" % k]))
        if "Mn_%d" % k in ops:
            Mn = np.ascontiguousarray(refM).copy()
            o.acfo_grad_mag_norm(F(Mn), F(np.ascontiguousarray(ops["S_%d" % k])), h, w, np.float32(0.005))
            refMn = ops["Mn_%d" % k]
            assert np.all(np.abs(Mn - refMn) <= 1.1 * RCP * np.abs(refMn) + 1e-12)


@pytest.mark.parametrize("name", PIPES)
def test_oracle_matches_pipeline_golden(oracle, name):
    z, model = load_pipeline(name)
    H, W, d_in = int(z["H"]), int(z["W"]), int(z["d_in"])
    plan = oracle.Plan(model, H, W, d_in)
    assert plan.nScales == len(z["scales"])
    lv = plan.levels
    assert np.array_equal(np.asarray([lv[i].scale for i in range(plan.nScales)]), z["scales"])
    geom = np.asarray([(lv[i].isReal, lv[i].realIndex, lv[i].hC, lv[i].wC, lv[i].hP, lv[i].wP, lv[i].nWinR, lv[i].nWinC, lv[i].offset)
                       for i in range(plan.nScales)], dtype=np.int64)
    assert np.array_equal(geom, z["level_geom"])
    pyr, _, _ = oracle.chns_pyramid(plan, z["frame"])
    assert np.array_equal(bits(pyr), bits(z["pyramid"]))
    if "ldcf_pyramid" in z.files:
        lvL, pyrL, _ = oracle.ldcf(plan, pyr)
        assert np.array_equal(bits(pyrL), bits(z["ldcf_pyramid"]))
        g = np.asarray([(lvL[i].hP, lvL[i].wP, lvL[i].nWinR, lvL[i].nWinC, lvL[i].offset) for i in range(plan.nScales)], dtype=np.int64)
        assert np.array_equal(g, z["ldcf_geom"])
        det, hits = oracle.detect_ldcf(plan, lvL, pyrL)
    else:
        det, hits = oracle.detect(plan, pyr)
    assert det.tobytes() == z["det"].tobytes() and hits.tobytes() == z["hits"].tobytes()


def test_get_scales_golden():
    """The product's host-side getScales (C ABI, no device) against the frozen lists."""
    lib = capi.load()
    z = np.load(os.path.join(GOLD, "scales.npz"))
    for k, (H, W, npo, nou, mh, mw, sh) in enumerate(z["cfgs"]):
        s = (C.c_double * 256)()
        a = (C.c_double * 256)()
        b = (C.c_double * 256)()
        n = C.c_int()
        assert lib.acf_hip_get_scales(int(npo), int(nou), int(mh), int(mw), int(sh), int(H), int(W), s, a, b, 256, C.byref(n)) == 0
        assert n.value == len(z["scales_%d" % k])
        assert np.array_equal(np.asarray(s[:n.value]), z["scales_%d" % k])
        assert np.array_equal(np.asarray(a[:n.value]), z["shw_h_%d" % k])
        assert np.array_equal(np.asarray(b[:n.value]), z["shw_w_%d" % k])
    # headline config: 31 scales, 4 real ones 1.0 / 0.5 / ~0.252 / ~0.1253
    s1 = z["scales_1"]
    assert len(s1) == 31 and s1[0] == 1.0 and s1[8] == 0.5


# ------------------------------------------------------------------ GPU: HIP path vs golden

@pytest.fixture(scope="module")
def dev():
    from acf_amd.detector import HipDetector
    d = HipDetector()
    yield d
    d.close()


@pytest.mark.gpu
def test_hip_ops_match_reference_kernel_outputs(dev, ops):
    for k, (h, w) in enumerate(ops["sizes"]):
        h, w = int(h), int(w)
        a = np.ascontiguousarray(ops["in%d" % k])
        assert np.array_equal(bits(dev.op_conv_tri(a, 1.0, aliased=False)), bits(ops["tri1_%d" % k]))
        assert np.array_equal(bits(dev.op_conv_tri(a, 1.0, aliased=True)), bits(ops["tri1_aliased_%d" % k]))
        for rad in (5, 2):
            key = "tri_r%d_%d" % (rad, k)
            if key in ops:
                assert np.array_equal(bits(dev.op_conv_tri(a[:1], float(rad))[0]), bits(ops[key])), key
        M, O, _ = dev.op_gradient_mag(a[0])
        refM, refO = ops["M_%d" % k], ops["O_%d" % k]
        assert np.all(np.abs(M - refM) <= 2.2 * RCP * np.abs(refM) + 1e-12)
        # orientation: table lookup of a ~1e-4-quantised cosine; away from flat pixels the two agree to a few table steps
        strong = refM > 1e-2
        assert np.all(np.abs(O - refO)[strong] <= 0.05)
        if "H_%d" % k in ops:
            H = dev.op_gradient_hist(refM, refO, 4, 6, 0)
            assert np.array_equal(bits(H), bits(ops["H_%d" % k]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", PIPES)
def test_hip_pipeline_matches_golden(name):
    import torch
    from acf_amd.detector import HipDetector
    z, model = load_pipeline(name)
    H, W, d_in = int(z["H"]), int(z["W"]), int(z["d_in"])
    det = HipDetector(model, H, W, d_in, max_batch=2, max_hits=1 << 14)
    fr = torch.from_numpy(np.stack([z["frame"], z["frame"]])).cuda()
    det.run(fr)
    for f in (0, 1):
        assert np.array_equal(bits(det.read_pyramid(f)), bits(z["pyramid"]))
        if "ldcf_pyramid" in z.files:
            from acf_amd import capi
            nCk = det.nChns * int(model["ldcfK"])
            got = np.concatenate([det.read_tap(f, capi.TAP_LDCF, i, (nCk, int(g[1]), int(g[0]))).ravel() for i, g in enumerate(z["ldcf_geom"])])
            assert np.array_equal(bits(got), bits(z["ldcf_pyramid"]))
        d, h = det.detections(f)
        assert d.tobytes() == z["det"].tobytes() and h.tobytes() == z["hits"].tobytes()
    det.close()


def test_ldcf_oracle_properties(oracle):
    """LDCF has no reference vectors (README.rst:8): pin the oracle's restatement through properties of the definition —
    a centre-tap filter is the identity, a one-tap filter is a zero-filled shift in the direction of a true convolution
    (conv2, not correlation), the operator is linear in the filter, and levels are sized round(.5 * size)."""
    import ctypes as C
    from acf_amd import capi, synth
    lib = oracle.lib()
    lib.acfo_ldcf_conv.argtypes = [oracle.fp, oracle.fp, C.c_int, C.c_int, oracle.fp]
    lib.acfo_ldcf_conv.restype = None
    h, w = 11, 7
    a = synth.uniform(9, h * w, 0).astype(np.float32).reshape(w, h)

    def conv(f):
        out = np.zeros_like(a)
        f = np.ascontiguousarray(f, np.float32)  # [dx][dy]
        lib.acfo_ldcf_conv(oracle.F(a), oracle.F(out), h, w, oracle.F(f))
        return out

    ident = np.zeros((5, 5), np.float32)
    ident[2, 2] = 1
    assert np.array_equal(conv(ident), a)
    one = np.zeros((5, 5), np.float32)
    one[2 + 1, 2 - 2] = 1          # tap (dx = +1, dy = -2): out(y, x) = in(y + 2, x - 1)
    want = np.zeros_like(a)
    want[1:, :h - 2] = a[:w - 1, 2:]
    assert np.array_equal(conv(one), want)
    f1 = (synth.uniform(4, 25, 1).reshape(5, 5) - 0.5).astype(np.float32)
    assert np.allclose(conv(f1) + conv(ident), conv(f1 + ident), atol=1e-6)
    # level sizes and window grid of the halved pyramid
    m = synth.make_model(seed=3, name="TINY", nTrees=8, ldcfK=2, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32, minDs_h=32, minDs_w=32)
    plan = oracle.Plan(m, 131, 175, 3)
    lvL = (capi.Level * len(plan.levels))()
    tot = lib.acfo_ldcf_plan(C.byref(plan.params), plan.levels, plan.nScales, plan.nChns, lvL)
    acc = 0
    for i in range(plan.nScales):
        assert lvL[i].hP == int(np.floor(0.5 * plan.levels[i].hP + 0.5)) and lvL[i].wP == int(np.floor(0.5 * plan.levels[i].wP + 0.5))
        assert lvL[i].nWinR == max(0, int(np.ceil((lvL[i].hP * 8 - 32 + 1) / 4.0)))
        assert lvL[i].offset == acc
        acc += plan.nChns * 2 * lvL[i].hP * lvL[i].wP
    assert tot == acc


# ------------------------------------------------------------------ reference bytes of resample / rgbConvert (round 3)

def _rnd77(seed, shape):
    n = int(np.prod(shape))
    return synth.uniform(seed, n, 77).astype(np.float32).reshape(shape)  # tests/golden/make_golden.py: rnd


@pytest.fixture(scope="module")
def rsl():
    return np.load(os.path.join(GOLD, "ref_resample_luv.npz"))


def test_oracle_matches_reference_resample_and_luv_bytes(oracle, rsl):
    """CPU leg: the restatement against the frozen outputs of the reference's own resample / rgbConvert bodies (the GPU box
    has no /root/reference; where oracle/_ref is present tests/test_oracle_vs_ref.py repeats this on many more cases)."""
    for k, (ha, wa, hb, wb, d, g) in enumerate(rsl["resample_cases"]):
        ha, wa, hb, wb, d = int(ha), int(wa), int(hb), int(wb), int(d)
        a = np.ascontiguousarray(_rnd77(500 + k, (d, wa, ha)))
        out = np.zeros((d, wb, hb), np.float32)
        assert oracle.lib().acfo_resample(oracle.F(a), oracle.F(out), ha, hb, wa, wb, d, np.float32(g)) == 0
        assert np.array_equal(bits(out), bits(rsl["rs_out%d" % k])), k
    for k, (h, w) in enumerate(rsl["luv_sizes"]):
        h, w = int(h), int(w)
        a = np.ascontiguousarray(synth.make_frame(600 + k, h, w, "rgb"))
        luv, gray = np.zeros((3, w, h), np.float32), np.zeros((w, h), np.float32)
        oracle.lib().acfo_rgb2luv(oracle.F(a), oracle.F(luv), h * w)
        oracle.lib().acfo_rgb2gray(oracle.F(a), oracle.F(gray), h * w)
        ref = rsl["luv_out%d" % k]
        assert np.array_equal(bits(gray), bits(rsl["gray_out%d" % k]))
        assert np.array_equal(bits(luv[0]), bits(ref[0]))
        if (h * w) % 4:
            assert np.array_equal(bits(luv), bits(ref))
        else:
            assert np.abs(luv - ref).max() <= 2 * RCP


@pytest.mark.gpu
def test_hip_resample_matches_reference_bytes(dev, rsl):
    """k_resample / k_resample_tile (acf_hip_op_im_resample) against the reference's own `resample` output, bit for bit."""
    for k, (ha, wa, hb, wb, d, g) in enumerate(rsl["resample_cases"]):
        ha, wa, hb, wb, d = int(ha), int(wa), int(hb), int(wb), int(d)
        a = np.ascontiguousarray(_rnd77(500 + k, (d, wa, ha)))
        got = dev.op_im_resample(a, hb, wb, float(np.float32(g)))
        assert np.array_equal(bits(got), bits(rsl["rs_out%d" % k])), (k, ha, wa, hb, wb)


@pytest.mark.gpu
def test_hip_rgb_convert_matches_reference_bytes(dev, rsl):
    """k_rgb2luv / k_rgb2gray against the reference's rgbConvert: gray and L bit-exact; U, V bit-exact where the reference
    takes its scalar body (n % 4 != 0) and within its one _mm_rcp_ps where it takes rgb2luv_sse."""
    from acf_amd import capi
    for k, (h, w) in enumerate(rsl["luv_sizes"]):
        h, w = int(h), int(w)
        a = np.ascontiguousarray(synth.make_frame(600 + k, h, w, "rgb"))
        ref = rsl["luv_out%d" % k]
        luv = dev.op_rgb_convert(a, capi.CS_LUV)
        gray = dev.op_rgb_convert(a, capi.CS_GRAY)
        assert np.array_equal(bits(gray.reshape(w, h)), bits(rsl["gray_out%d" % k]))
        assert np.array_equal(bits(luv[0]), bits(ref[0]))
        if (h * w) % 4:
            assert np.array_equal(bits(luv), bits(ref)), k
        else:
            l = ref[0].astype(np.float64)
            for ch, c, mn in ((1, 13 * 0.197833, -88.0 / 270), (2, 13 * 0.468331, -134.0 / 270)):
                mag = np.abs(ref[ch].astype(np.float64) + mn + c * l)
                assert (np.abs(luv[ch].astype(np.float64) - ref[ch]) <= 1.1 * RCP * mag + 4e-7).all(), (k, ch)
