"""GPU parity, operator by operator: each HIP kernel (called through the C ABI's
acf_hip_op_* entry points) against the oracle on the same seeded inputs.
Everything here is bit-exact: the kernels and the oracle evaluate the same IEEE
expressions in the same order."""
import ctypes as C

import numpy as np
import pytest

from acf_amd import capi, synth

pytestmark = pytest.mark.gpu

SIZES = [(64, 48), (63, 50), (37, 41), (120, 160), (270, 480), (1080, 192)]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rnd(seed, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * synth.uniform(seed, n, 3)).astype(np.float32).reshape(shape)


@pytest.fixture(scope="module")
def dev():
    from acf_amd.detector import HipDetector
    d = HipDetector()
    yield d
    d.close()


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("aliased", [True, False])
def test_conv_tri1(dev, oracle, h, w, aliased):
    a = rnd(h * 13 + w, (3, w, h))
    got = dev.op_conv_tri(a, 1.0, aliased=aliased)
    want = a.copy()
    if aliased:
        assert oracle.lib().acfo_conv_tri1(oracle.F(want), oracle.F(want), h, w, 3, 2.0, 1) == 0
    else:
        src = a.copy()
        assert oracle.lib().acfo_conv_tri1(oracle.F(src), oracle.F(want), h, w, 3, 2.0, 1) == 0
    assert np.array_equal(bits(got), bits(want))


def test_conv_tri1_tall_plane(dev, oracle):
    """R = 3 rows per thread (taller than 2048 rows: the 4K case)."""
    h, w = 2160, 24
    a = rnd(5, (1, w, h))
    got = dev.op_conv_tri(a, 1.0, aliased=True)
    want = a.copy()
    oracle.lib().acfo_conv_tri1(oracle.F(want), oracle.F(want), h, w, 1, 2.0, 1)
    assert np.array_equal(bits(got), bits(want))


def test_conv_tri1_other_radius(dev, oracle):
    h, w = 96, 72
    a = rnd(6, (2, w, h))
    r = 0.5
    p = np.float32(12.0 / r / (r + 2.0) - 2.0)
    got = dev.op_conv_tri(a, r, aliased=True)
    want = a.copy()
    oracle.lib().acfo_conv_tri1(oracle.F(want), oracle.F(want), h, w, 2, float(p), 1)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("r", [5, 2])
def test_conv_tri_running_sums(dev, oracle, h, w, r):
    a = rnd(h * 3 + w * 5 + r, (2, w, h))
    got = dev.op_conv_tri(a, float(r), aliased=False)
    want = np.zeros_like(a)
    assert oracle.lib().acfo_conv_tri(oracle.F(a), oracle.F(want), h, w, 2, r, 1) == 0
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("h", [44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 92, 96, 100, 104, 124, 128, 132, 136, 156, 160, 164, 272, 540])
def test_conv_tri_r5_streaming_boundaries(dev, oracle, h):
    """Every residue of the 16-row unrolled streaming column kernel (k_tri_y5) and its head / tail hand-over."""
    w = 70
    a = rnd(h * 7 + 1, (1, w, h))
    got = dev.op_conv_tri(a, 5.0, aliased=False)
    want = np.zeros_like(a)
    assert oracle.lib().acfo_conv_tri(oracle.F(a), oracle.F(want), h, w, 1, 5, 1) == 0
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("w", [44, 48, 49, 55, 56, 63, 64, 65, 71, 72, 79, 80, 81, 95, 96, 97, 112, 113, 480])
def test_conv_tri_r5_x_streaming_boundaries(dev, oracle, w):
    """Every residue of the 16-column unrolled, two-set x-pass kernel (k_tri_x5v) and its head / tail hand-over."""
    h = 64
    a = rnd(w * 11 + 3, (1, w, h))
    got = dev.op_conv_tri(a, 5.0, aliased=False)
    want = np.zeros_like(a)
    assert oracle.lib().acfo_conv_tri(oracle.F(a), oracle.F(want), h, w, 1, 5, 1) == 0
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("full", [0, 1])
def test_gradient_mag(dev, oracle, h, w, full):
    a = synth.make_frame(h + w, h, w, "gray")[0]
    M, O, _ = dev.op_gradient_mag(a, 0, 0.005, full)
    Mo, Oo = np.zeros_like(a), np.zeros_like(a)
    assert oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(Mo), oracle.F(Oo), h, w, 1, full) == 0
    assert np.array_equal(bits(M), bits(Mo))
    assert np.array_equal(bits(O), bits(Oo))


def test_gradient_mag_flat_and_extreme(dev, oracle):
    """M2 == 0 (rsqrt -> inf, clamped to 1e10), exact +-1 cosines, denormal gradients."""
    h, w = 32, 32
    a = np.zeros((w, h), np.float32)
    a[8:16, :] = 1.0           # pure Gx edges
    a[:, 20:] += 0.5           # pure Gy edges
    a[24:, :] += np.float32(1e-39)  # denormal step
    M, O, _ = dev.op_gradient_mag(a, 0, 0.005, 0)
    Mo, Oo = np.zeros_like(a), np.zeros_like(a)
    oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(Mo), oracle.F(Oo), h, w, 1, 0)
    assert np.array_equal(bits(M), bits(Mo))
    assert np.array_equal(bits(O), bits(Oo))
    assert M.min() == np.float32(1.0) / np.float32(1e10)


@pytest.mark.parametrize("h,w", [(64, 48), (120, 160), (272, 484)])
def test_gradient_mag_normalised(dev, oracle, h, w):
    a = synth.make_frame(h * w, h, w, "gray")[0]
    M, O, S = dev.op_gradient_mag(a, 5, 0.005, 0)
    Mo, Oo, So = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
    oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(Mo), oracle.F(Oo), h, w, 1, 0)
    oracle.lib().acfo_conv_tri(oracle.F(Mo), oracle.F(So), h, w, 1, 5, 1)
    oracle.lib().acfo_grad_mag_norm(oracle.F(Mo), oracle.F(So), h, w, 0.005)
    assert np.array_equal(bits(S), bits(So))
    assert np.array_equal(bits(M), bits(Mo))


@pytest.mark.parametrize("h,w", [(64, 48), (120, 160), (1080, 64)])
@pytest.mark.parametrize("bin", [4, 2])
@pytest.mark.parametrize("full", [0, 1])
@pytest.mark.parametrize("softBin", [0, 2, -2])
def test_gradient_hist(dev, oracle, h, w, bin, full, softBin):
    M = rnd(h + w + bin, (w, h), 0, 0.6)
    hi = 2 * np.pi if full else np.pi
    O = rnd(h + w + bin + 1, (w, h), 0.0, float(hi) - 1e-6)
    got = dev.op_gradient_hist(M, O, bin, 6, full, softBin)
    want = np.zeros_like(got)
    assert oracle.lib().acfo_grad_hist(oracle.F(M), oracle.F(O), oracle.F(want), h, w, bin, 6, softBin, full) == 0
    assert np.array_equal(bits(got), bits(want))
    if softBin < 0:
        soft = np.zeros_like(got)
        assert oracle.lib().acfo_grad_hist(oracle.F(M), oracle.F(O), oracle.F(soft), h, w, bin, 6, 0, full) == 0
        assert not np.array_equal(bits(soft), bits(want))  # (the two branches do differ on this input)


RESAMPLE_CASES = [
    # (ha, wa, hb, wb): exact /2 /3 /4, mixed exact, generic down (2,3,4,>4 taps), up, up x2, down in one axis / up in other
    (64, 48, 32, 24), (60, 48, 20, 16), (64, 48, 16, 12), (64, 48, 32, 12),
    (100, 80, 71, 57), (100, 80, 51, 41), (135, 240, 124, 220), (540, 960, 136, 240), (540, 96, 68, 12),
    (64, 48, 91, 67), (48, 64, 96, 128), (68, 121, 62, 110), (30, 40, 33, 37), (270, 480, 248, 441),
    (540, 960, 272, 484), (400, 90, 100, 18), (333, 77, 65, 70), (130, 700, 17, 60),
    (480, 640, 960, 1280), (61, 47, 100, 64), (50, 30, 52, 31), (7, 9, 28, 36), (120, 33, 124, 66),  # both axes up, whole quads of rows: k_resample_up (x2, odd ratios, barely up, x4)  # LDS-tiled path: ~2x, exact /4 tall, 5+ taps in y with up in x (generic), 7+ taps
]


@pytest.mark.parametrize("ha,wa,hb,wb", RESAMPLE_CASES)
@pytest.mark.parametrize("nrm", [1.0, 1.0832])
def test_im_resample(dev, oracle, ha, wa, hb, wb, nrm):
    a = rnd(ha * 7 + wb, (2, wa, ha))
    got = dev.op_im_resample(a, hb, wb, nrm)
    want = np.zeros_like(got)
    assert oracle.lib().acfo_resample(oracle.F(a), oracle.F(want), ha, hb, wa, wb, 2, np.float32(nrm)) == 0
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("h,w", [(64, 48), (63, 49), (120, 160)])
def test_rgb2luv(dev, oracle, h, w):
    a = synth.make_frame(h * w + 1, h, w, "rgb")
    got = dev.op_rgb_convert(a, capi.CS_LUV)
    want = np.zeros_like(a)
    oracle.lib().acfo_rgb2luv(oracle.F(a), oracle.F(want), h * w)
    assert np.array_equal(bits(got), bits(want))
    g = dev.op_rgb_convert(a, capi.CS_GRAY)
    wg = np.zeros((1, w, h), np.float32)
    oracle.lib().acfo_rgb2gray(oracle.F(a), oracle.F(wg), h * w)
    assert np.array_equal(bits(g), bits(wg))
    a[:, :3, :5] = np.float32(0.5)  # grey pixels: rgb2hsv's first branch
    a[0, 3:6, :5] = a[1, 3:6, :5]   # r == g ties
    hsv = dev.op_rgb_convert(a, capi.CS_HSV)
    wh = np.zeros_like(a)
    oracle.lib().acfo_rgb2hsv(oracle.F(a), oracle.F(wh), h * w)
    assert np.array_equal(bits(hsv), bits(wh))


@pytest.mark.parametrize("tiles", [1, 0])
@pytest.mark.parametrize("depth", [2, 1, 3, 0])
@pytest.mark.parametrize("nTrees", [128, 300, 20, 40, 70])
def test_acf_detect1(dev, oracle, depth, nTrees, tiles):
    """The cascade on a random channel buffer: hits identical in number, order, position and score bits.
    tiles=1: the LDS-tiled paths — depth 2: k_cascade_tile2 (its stages end at 16/32/64/128 trees, then the wave-per-window
    tail); depths 1 and 3: k_cascade_tileD for trees [0, 32) (or the whole model when it has <= 32 trees: 20 trees at depth 3 /
    20 at depth 1 are a multiple of its batch, so the survivors are hits), the staged queue / tail kernels behind it;
    tiles=0 and depth 0: the global-memory staged path (first / queue / LDS tail kernels)."""
    if tiles and depth == 0:
        pytest.skip("variable-depth trees (child walk): staged path only")
    nC, wP, hP = 10, 60, 44
    chns = rnd(99 + depth, (nC, wP, hP), 0.0, 0.6)
    kw = dict(treeDepth=depth)
    m = synth.make_model(seed=11 + depth, name="TINY", nTrees=nTrees, cascThr=-1.0 if nTrees < 300 else -4.0, **kw)
    # thresholds inside the data range so both branches are taken; leaf values with a
    # slight negative drift so that part of the windows is rejected at every depth
    m["thrs"] = rnd(5, m["thrs"].shape, 0.1, 0.5)
    m["hs"] = rnd(6, m["hs"].shape, -0.25, 0.2)
    m["fids"] = (synth.uniform(7, m["fids"].size, 1) * (nC * 16)).astype(np.uint32).reshape(m["fids"].shape)
    dev.set_option("cascade_tiles", tiles)
    dev.set_model(m)
    try:
        got = dev.op_acf_detect1(chns)
    finally:
        dev.set_option("cascade_tiles", 1)
    params, keep = capi.make_params(m)
    want = np.zeros(1 << 16, dtype=capi.HIT_DTYPE)
    n = oracle.lib().acfo_acf_detect1(chns.ctypes.data_as(C.c_void_p), 0, keep["thrs"].ctypes.data_as(C.c_void_p), hP, wP, nC,
                                      C.byref(params), want.ctypes.data_as(C.POINTER(capi.Hit)), 1 << 16, 0)
    want = want[:n]
    assert n > 0 and n < (wP - 3) * (hP - 3), n
    assert len(got) == n
    for k in ("scale", "c", "r"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(bits(got["score"]), bits(want["score"]))


def _detect1_case(dev, oracle, depth, nTrees, tiles, cascThr, hs_range, seed=0):
    nC, wP, hP = 10, 60, 44
    chns = rnd(99 + depth + seed, (nC, wP, hP), 0.0, 0.6)
    m = synth.make_model(seed=11 + depth, name="TINY", nTrees=nTrees, cascThr=cascThr, treeDepth=depth)
    m["thrs"] = rnd(5, m["thrs"].shape, 0.1, 0.5)
    m["hs"] = rnd(6, m["hs"].shape, *hs_range)
    m["fids"] = (synth.uniform(7, m["fids"].size, 1) * (nC * 16)).astype(np.uint32).reshape(m["fids"].shape)
    dev.set_option("cascade_tiles", tiles)
    dev.set_model(m)
    try:
        got = dev.op_acf_detect1(chns)
    finally:
        dev.set_option("cascade_tiles", 1)
    params, keep = capi.make_params(m)
    want = np.zeros(1 << 16, dtype=capi.HIT_DTYPE)
    n = oracle.lib().acfo_acf_detect1(chns.ctypes.data_as(C.c_void_p), 0, keep["thrs"].ctypes.data_as(C.c_void_p), hP, wP, nC,
                                      C.byref(params), want.ctypes.data_as(C.POINTER(capi.Hit)), 1 << 16, 0)
    want = want[:n]
    assert 0 < n < (wP - 3) * (hP - 3), n
    assert len(got) == n
    for k in ("scale", "c", "r"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(bits(got["score"]), bits(want["score"]))


@pytest.mark.parametrize("nTrees", [70, 300])
@pytest.mark.parametrize("depth", [4, 5, 6, 7, 8])
def test_acf_detect1_deep_trees(dev, oracle, depth, nTrees):
    """acfDetect1.cpp:184-228 dispatches treeDepth 1..8 (getChild walks depth levels of a full binary tree of
    2^(depth+1)-1 nodes): every depth above the depth-2 fast path runs the staged global-memory cascade."""
    _detect1_case(dev, oracle, depth, nTrees, 1, -1.0 if nTrees < 300 else -4.0, (-0.25, 0.2))


@pytest.mark.parametrize("depth,tiles", [(2, 1), (2, 0), (5, 1), (0, 1), (3, 1), (1, 1), (4, 1)])
def test_acf_detect1_4096_trees(dev, oracle, depth, tiles):
    """4096 trees (the largest detectors of the toolbox): the tile kernel's stage E and k_tail_scan then walk 3968 tail trees
    (62 batches of 64: more than four per wave), the staged path its long last stage.  Zero-mean leaves: the score is a
    random walk, cascThr -6 lets a small fraction through."""
    _detect1_case(dev, oracle, depth, 4096, tiles, -6.0, (-0.2, 0.2), seed=3)


@pytest.mark.parametrize("depth", [2, 0, 3])
def test_evaluate_single_window(dev, oracle, depth):
    """Detector::evaluate (acfDetect1.cpp:337-342): the score of the window at (0,0) with cascThr = 0 — the value at which
    the tree loop stops, also for windows that are not detections."""
    nC, wP, hP = 10, 9, 7
    m = synth.make_model(seed=11 + depth, name="TINY", nTrees=128, cascThr=-1.0, treeDepth=depth)
    m["thrs"] = rnd(5, m["thrs"].shape, 0.1, 0.5)
    m["fids"] = (synth.uniform(7, m["fids"].size, 1) * (nC * 16)).astype(np.uint32).reshape(m["fids"].shape)
    dev.set_model(m)
    params, keep = capi.make_params(m)
    seen = set()
    for k, (lo, hi) in enumerate([(-0.25, 0.2), (-0.05, 0.3), (0.0, 0.3), (-0.3, 0.0)]):
        m["hs"] = rnd(6 + k, m["hs"].shape, lo, hi)
        dev.set_model(m)
        params, keep = capi.make_params(m)
        chns = rnd(50 + k, (nC, wP, hP), 0.0, 0.6)
        got = dev.op_evaluate(chns)
        want = oracle.lib().acfo_evaluate(oracle.F(chns), hP, wP, nC, C.byref(params), np.float32(0.0))
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (k, got, want)
        seen.add(bool(want > 0))
    assert seen == {True, False}  # both an early stop and a full pass were exercised


@pytest.mark.parametrize("explicit_thrs", [False, True])
@pytest.mark.parametrize("depth", [2, 0, 3])
def test_acf_detect1_u8(dev, oracle, depth, explicit_thrs):
    """The uint8_t cascade body (acfDetect1.cpp:157-166,187-192): byte planes + Classifier::thrsU8.  Ties between a
    byte feature and a byte threshold are common here (256 levels), so `<` vs `<=` mistakes cannot hide."""
    nC, wP, hP = 10, 60, 44
    chns = (rnd(77 + depth, (nC, wP, hP), 0.0, 0.6) * 255.0).astype(np.uint8)
    m = synth.make_model(seed=11 + depth, name="TINY", nTrees=128, cascThr=-1.0, treeDepth=depth)
    m["thrs"] = rnd(5, m["thrs"].shape, 0.1, 0.5)
    m["hs"] = rnd(6, m["hs"].shape, -0.25, 0.2)
    m["fids"] = (synth.uniform(7, m["fids"].size, 1) * (nC * 16)).astype(np.uint32).reshape(m["fids"].shape)
    dev.set_model(m)
    tu = oracle.thrs_u8(m["thrs"])
    got = dev.op_acf_detect1_u8(chns, tu if explicit_thrs else None)
    params, keep = capi.make_params(m)
    want = np.zeros(1 << 16, dtype=capi.HIT_DTYPE)
    n = oracle.lib().acfo_acf_detect1(chns.ctypes.data_as(C.c_void_p), 1, tu.ctypes.data_as(C.c_void_p), hP, wP, nC,
                                      C.byref(params), want.ctypes.data_as(C.POINTER(capi.Hit)), 1 << 16, 0)
    want = want[:n]
    assert n > 0 and n < (wP - 3) * (hP - 3), n
    assert len(got) == n
    for k in ("scale", "c", "r"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(bits(got["score"]), bits(want["score"]))


@pytest.mark.parametrize("case", ["stride8", "wide_model", "tall_plane", "one_window", "permissive", "stride2", "stride1", "stride2_odd", "stride3"])
def test_acf_detect1_tiled_geometries(dev, oracle, case):
    """Tile edge cases of the LDS-tiled cascade: step 2 between windows, a non-square model, a plane with
    many tiles along r, a plane that holds exactly one window, and a threshold nothing is rejected by
    (every window reaches the tail queue).  stride < shrink (acfDetect1.cpp:88-96: window (r, c) sits at cell
    r * stride / shrink, so shrink / stride windows share an offset): the cascade runs once per distinct offset and
    k_expand_hits writes every window of it (stride 2 and 1 on cells of 4; a grid whose last offset has fewer windows;
    stride 3, which does not divide the shrink and keeps one evaluation per window)."""
    nC = 10
    kw = dict(name="TINY", nTrees=160, cascThr=-3.0)
    wP, hP = 70, 50
    if case == "stride8":
        kw.update(stride=8)
    elif case in ("stride2", "stride1", "stride3"):
        kw.update(stride=int(case[-1]), cascThr=-1.5)
        wP, hP = 40, 33
    elif case == "stride2_odd":
        kw.update(stride=2, cascThr=-1.5)
        wP, hP = 31, 28
    elif case == "wide_model":
        kw.update(modelDs_h=16, modelDs_w=40, modelDsPad_h=16, modelDsPad_w=40)
    elif case == "tall_plane":
        wP, hP = 24, 300
    elif case == "one_window":
        wP, hP = 4, 4
        kw.update(cascThr=-50.0)
    elif case == "permissive":
        wP, hP = 40, 45
        kw.update(cascThr=-50.0)
    m = synth.make_model(seed=21, **kw)
    mh, mw = m["modelDsPad_h"] // 4, m["modelDsPad_w"] // 4
    chns = rnd(123, (nC, wP, hP), 0.0, 0.6)
    m["thrs"] = rnd(5, m["thrs"].shape, 0.1, 0.5)
    m["hs"] = rnd(6, m["hs"].shape, -0.25, 0.2)
    m["fids"] = (synth.uniform(7, m["fids"].size, 1) * (nC * mh * mw)).astype(np.uint32).reshape(m["fids"].shape)
    dev.set_model(m)
    got = dev.op_acf_detect1(chns)
    params, keep = capi.make_params(m)
    want = np.zeros(1 << 16, dtype=capi.HIT_DTYPE)
    n = oracle.lib().acfo_acf_detect1(chns.ctypes.data_as(C.c_void_p), 0, keep["thrs"].ctypes.data_as(C.c_void_p), hP, wP, nC,
                                      C.byref(params), want.ctypes.data_as(C.POINTER(capi.Hit)), 1 << 16, 0)
    want = want[:n]
    assert n > 0, n
    assert len(got) == n, (len(got), n)
    for k in ("scale", "c", "r"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(bits(got["score"]), bits(want["score"]))


def test_gradmag_fast_reciprocals_equal_ieee_for_every_input():
    """k_grad_mag_vec / k_smooth_grad compute m = min(1/sqrt(m2), 1e10), M = 1/m with one v_rsq_f32 and FMA refinements
    (gm_inv_fast, kernels.hip.h) instead of the compiler's IEEE sqrt + two divisions.  The two forms are functions of ONE
    float, so the claim "same bits" is checked exhaustively on the device: every finite m2 >= 0 (2^31 - 2^23 patterns)."""
    import ctypes as C
    from acf_amd import capi
    lib = capi.load()
    ctx = C.c_void_p()
    assert lib.acf_hip_create(0, None, C.byref(ctx)) == 0
    bad, first = C.c_uint64(0), C.c_uint32(0)
    assert lib.acf_hip_selftest_gradmag(ctx, 0, 0x7f7fffff, C.byref(bad), C.byref(first)) == 0
    assert bad.value == 0, "%d mismatches, first at bits 0x%08x" % (bad.value, first.value)
    lib.acf_hip_destroy(ctx)
