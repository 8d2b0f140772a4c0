"""Option "arith" = 1 on the device: the reference's OWN arithmetic at its three approximate sites (gradMag, gradMagNorm,
rgb2luv_sse: T/gradientMex.cpp:209-219,266; T/rgbConvertMex.cpp:161; T/sse.hpp:185-192) from the build host's rcpps / rsqrtps
tables (tests/golden/x86_rcp_rsqrt.npz, checked against the instructions for all 2^32 inputs by make_x86_tables.py).

Everything here is compared with bytes the REFERENCE's compiled kernels produced on the build host (frozen in
tests/golden/ref_ops.npz, ref_resample_luv.npz, tref_study.npz), bit for bit — not within a bound:

 * the device's table functions == the oracle's for every one of the 2^32 inputs (digests);
 * acf_hip_op_gradient_mag's M, O and normalised M == the reference's gradMag / gradMagNorm bytes, incl. the odd-sized case
   whose last n % 4 elements take gradMagNorm's scalar division; acf_hip_op_rgb_convert == the reference's rgbConvert bytes
   (vector and scalar bodies);
 * the whole path at BASELINE.json's cfg 1 / 2 / 4 sizes: the cascade's hits == the hits of the reference's compiled kernels
   under the restated orchestration (`*_hits_ref`), scores included — the north star's "scores within 1e-4 of the reference"
   met with 0 difference on the reference's real SSE arithmetic;
 * the pyramid == the oracle's table-tier pyramid; switching the option off again restores the exact tier.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from acf_amd import capi, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dev():
    from acf_amd.detector import HipDetector
    from oracle import binding as ob
    d = HipDetector()
    d.set_x86_tables(*ob.x86_fixture())
    yield d
    d.close()


def test_option_needs_tables():
    from acf_amd.detector import HipDetector, HipError
    d = HipDetector()
    with pytest.raises(HipError):
        d.set_option("arith", 1)
    d.close()


def test_device_table_functions_equal_the_oracles_for_every_input(oracle, dev):
    oracle.set_x86_tables(*oracle.x86_fixture())
    chunks = 64
    per = (1 << 32) // chunks
    with ThreadPoolExecutor(min(32, os.cpu_count() or 4)) as ex:
        want = list(ex.map(lambda k: oracle.x86_digest(k * per, per, 1), range(chunks)))
    for k in range(chunks):
        got = dev.selftest_x86(k * per, per, 1)
        assert got[:2] == want[k], k
        assert got[2] == 0, k     # gradMag's one-read form (gm_inv_x86g) == rsqrt table, min, rcp table in turn, for every M2 >= 0 and NaN


def test_gradient_mag_and_norm_equal_the_reference_bytes(dev):
    ops = np.load(os.path.join(GOLD, "ref_ops.npz"))
    dev.set_option("arith", 1)
    try:
        for k, (h, w) in enumerate(ops["sizes"]):
            a = ops["in%d" % k][0]
            M, O, _ = dev.op_gradient_mag(a, 0)
            assert np.array_equal(bits(M), bits(ops["M_%d" % k])), k
            assert np.array_equal(bits(O), bits(ops["O_%d" % k])), k
            if "Mn_%d" % k in ops.files:
                Mn, _, S = dev.op_gradient_mag(a, 5, 0.005)
                assert np.array_equal(bits(S), bits(ops["S_%d" % k])), k
                assert np.array_equal(bits(Mn), bits(ops["Mn_%d" % k])), k   # (k = 2: n % 4 != 0, the scalar tail divides)
    finally:
        dev.set_option("arith", 0)
    # and the exact tier differs from those bytes (the option is not a no-op)
    M, _, _ = dev.op_gradient_mag(ops["in0"][0], 0)
    assert not np.array_equal(bits(M), bits(ops["M_0"]))


def test_rgb2luv_equals_the_reference_bytes(dev):
    rsl = np.load(os.path.join(GOLD, "ref_resample_luv.npz"))
    dev.set_option("arith", 1)
    try:
        for k, (h, w) in enumerate(rsl["luv_sizes"]):
            a = np.ascontiguousarray(synth.make_frame(600 + k, int(h), int(w), "rgb"))   # (make_golden.py's inputs)
            luv = dev.op_rgb_convert(a, capi.CS_LUV)
            assert np.array_equal(bits(luv), bits(rsl["luv_out%d" % k])), (k, h, w)
    finally:
        dev.set_option("arith", 0)


TREF_CFG = {
    "cfg1_vga_gray_face64": ("gray", "FACE64"),
    "cfg2_1080p_luv_face80": ("luv", "FACE80"),
    "cfg4_vga_rgb_inria": ("rgb", "INRIA"),
}


@pytest.mark.parametrize("cfg", list(TREF_CFG))
def test_hits_equal_the_reference_kernels_hits_bit_for_bit(oracle, cfg):
    import torch
    from acf_amd.detector import HipDetector
    fix = np.load(os.path.join(GOLD, "tref_study.npz"))
    H, W, d_in, nframes, seed0, mseed = [int(v) for v in fix[cfg + "_meta"]]
    kind, preset = TREF_CFG[cfg]
    model = synth.make_model(seed=mseed, name=preset)
    n = int(nframes)
    frames = np.stack([synth.make_frame(seed0 + f, H, W, kind) for f in range(n)])
    det = HipDetector(model, H, W, d_in, max_batch=n, max_hits=1 << 15)
    det.set_x86_tables(*oracle.x86_fixture())
    det.set_option("arith", 1)
    det.run(torch.from_numpy(frames).cuda())
    diffs = 0
    for f in range(n):
        _, gh = det.detections(f)
        want = fix["%s_f%d_hits_ref" % (cfg, f)]
        assert gh.tobytes() == want.tobytes(), (cfg, f, len(gh), len(want))
        diffs += int((gh["score"] != want["score"]).sum())
    assert diffs == 0
    # back to the exact tier on the same context: the T-exact side of the study
    det.set_option("arith", 0)
    det.run(torch.from_numpy(frames).cuda())
    for f in range(min(n, 2)):
        _, gh = det.detections(f)
        assert gh.tobytes() == fix["%s_f%d_hits_exact" % (cfg, f)].tobytes(), (cfg, f)
    det.close()


@pytest.mark.parametrize("shape", [("FACE80", 272, 480, "luv", 3), ("INRIA", 240, 320, "rgb", 3), ("FACE64", 240, 320, "gray", 1), ("TINY", 120, 150, "rgb", 3)])
def test_pyramid_equals_the_oracles_table_tier(oracle, shape):
    import torch
    from acf_amd.detector import HipDetector
    name, H, W, kind, d = shape
    model = synth.make_model(seed=3, name=name, nTrees=64)
    frames = np.stack([synth.make_frame(5 + f, H, W, kind) for f in range(2)])
    plan = oracle.Plan(model, H, W, d)
    oracle.set_x86_tables(*oracle.x86_fixture())
    det = HipDetector(model, H, W, d, max_batch=2, max_hits=1 << 15)
    det.set_x86_tables(*oracle.x86_fixture())
    det.set_option("arith", 1)
    det.pyramid(torch.from_numpy(frames).cuda())
    for f in range(2):
        oracle.set_approx(3)
        try:
            want, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_approx(0)
        got = det.read_pyramid(f)
        assert np.array_equal(bits(got), bits(want)), (shape, f)
        exact, _, _ = oracle.chns_pyramid(plan, frames[f])
        assert not np.array_equal(bits(got), bits(exact))
    det.close()


def test_u8_ingest_in_reference_arithmetic(oracle):
    """The packed 8-bit front end (k_ingest_u8 converts RGB -> LUV itself) under option "arith"."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 120, 160
    model = synth.make_model(seed=3, name="INRIA", nTrees=64)
    rgb = (synth.make_frame(11, H, W, "rgb") * 255.0 + 0.5).astype(np.uint8)     # [3][W][H]
    packed = np.ascontiguousarray(rgb.transpose(2, 1, 0))                          # [H][W][3]
    planar = oracle.aligned((3, W, H))
    oracle.lib().acfo_ingest_u8(packed.ctypes.data, H, W, 3, 0, 1, 2, W * 3, oracle.F(planar), 3)   # ACF.cpp:114-119,137
    plan = oracle.Plan(model, H, W, 3)
    oracle.set_x86_tables(*oracle.x86_fixture())
    oracle.set_approx(3)
    try:
        want, _, _ = oracle.chns_pyramid(plan, planar)
    finally:
        oracle.set_approx(0)
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 15)
    det.set_x86_tables(*oracle.x86_fixture())
    det.set_option("arith", 1)
    det.pyramid_u8(torch.from_numpy(packed[None]).cuda(), capi.PIX_RGB)
    got = det.read_pyramid(0)
    assert np.array_equal(bits(got), bits(want))
    det.close()


def test_reference_arithmetic_with_sub_batch_contexts_and_taps(oracle):
    """Option "streams" > 1 (sub-batch contexts share their parent's tables) and option "taps" (the unfused forms anyway) in the T-ref tier."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 240, 320
    model = synth.make_model(seed=3, name="INRIA", nTrees=64)
    frames = np.stack([synth.make_frame(15 + f, H, W, "rgb") for f in range(6)])
    plan = oracle.Plan(model, H, W, 3)
    oracle.set_x86_tables(*oracle.x86_fixture())
    want = []
    for f in range(6):
        oracle.set_approx(3)
        try:
            pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_approx(0)
        want.append((pyr, oracle.detect(plan, pyr)))
    for kw in (dict(streams=2), dict(taps=True)):
        det = HipDetector(streams=kw.get("streams", 1), taps=kw.get("taps", False))
        det.set_x86_tables(*oracle.x86_fixture())     # before the plan: the children are created by it
        det.set_option("arith", 1)
        det.set_model(model)
        det.plan(H, W, 3, max_batch=6, max_hits=1 << 15)
        det.run(torch.from_numpy(frames).cuda())
        for f in range(6):
            d, h = det.detections(f)
            assert d.tobytes() == want[f][1][0].tobytes() and h.tobytes() == want[f][1][1].tobytes(), (kw, f)
            if "taps" in kw:
                assert np.array_equal(bits(det.read_pyramid(f)), bits(want[f][0])), f
        det.close()
    # tables installed AFTER the plan reach the children too
    det = HipDetector(model, H, W, 3, max_batch=6, max_hits=1 << 15, streams=2)
    det.set_x86_tables(*oracle.x86_fixture())
    det.set_option("arith", 1)
    det.run(torch.from_numpy(frames).cuda())
    for f in range(6):
        d, h = det.detections(f)
        assert d.tobytes() == want[f][1][0].tobytes() and h.tobytes() == want[f][1][1].tobytes(), f
    det.close()


@pytest.mark.parametrize("cfg", list(TREF_CFG))
def test_hits_equal_the_reference_kernels_running_on_this_host(oracle, cfg):
    """The strongest form of the parity claim, LIVE on the box the test runs on: the reference's own compiled SSE kernels
    (oracle/_ref/libacfref.so: convTri1, convTri, gradMag, gradMagNorm, gradHist, resample, rgbConvert — built in the build container
    from the sources where they lie, shipped as a binary) execute on THIS host's CPU under the restated orchestration, and the HIP path,
    with the tables probed from the same CPU (what acf::HipDetector::setReferenceArithmetic(true) installs), must give the same hits,
    positions and score bits — on an Intel build host or an AMD EPYC GPU box alike.  Skips where the reference binary is absent or
    the CPU's rcpps / rsqrtps are not functions of the top 12 mantissa bits."""
    import torch
    from acf_amd.detector import HipDetector
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libacfref.so not shipped")
    t = oracle.x86_probe()
    if t is None:
        pytest.skip("not an SSE host")
    oracle.set_x86_tables(*t)
    if oracle.x86_verify(0, 1 << 24, 255) != (0, 0) or oracle.x86_verify(0x3f800000, 1 << 23, 1) != (0, 0) or oracle.x86_verify(0x7f000000, 1 << 24, 1) != (0, 0):
        pytest.skip("this CPU's rcpps / rsqrtps are not table functions of the top 12 mantissa bits")
    fix = np.load(os.path.join(GOLD, "tref_study.npz"))
    H, W, d_in, nframes, seed0, mseed = [int(v) for v in fix[cfg + "_meta"]]
    kind, preset = TREF_CFG[cfg]
    model = synth.make_model(seed=mseed, name=preset)
    n = 3 if H > 600 else 6
    frames = np.stack([synth.make_frame(seed0 + 100 + f, H, W, kind) for f in range(n)])   # (other frames than the fixture's)
    plan = oracle.Plan(model, H, W, d_in)
    det = HipDetector(model, H, W, d_in, max_batch=n, max_hits=1 << 15)
    det.set_x86_tables(*t)
    det.set_option("arith", 1)
    det.run(torch.from_numpy(frames).cuda())
    total = 0
    for f in range(n):
        oracle.set_tref(True)                        # every toolbox kernel = the reference's compiled one, on this CPU
        try:
            pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_tref(False)
        _, want = oracle.detect(plan, pyr)
        _, gh = det.detections(f)
        assert gh.tobytes() == want.tobytes(), (cfg, f, len(gh), len(want))
        if f == 0:
            det.set_option("keep_pyramid", 1)
        total += len(want)
    assert total > 0
    # and the pyramid itself, level by level, for one frame
    det.pyramid(torch.from_numpy(frames[:1]).cuda())
    oracle.set_tref(True)
    try:
        pyr, _, _ = oracle.chns_pyramid(plan, frames[0])
    finally:
        oracle.set_tref(False)
    assert np.array_equal(bits(det.read_pyramid(0)), bits(pyr))
    det.close()


@pytest.mark.parametrize("name,kind,kw", [("luv_tiny_160x120", "luv", dict(name="TINY", nTrees=64, cascThr=-3.0)),
                                          ("rgb_inria_160x120", "rgb", dict(name="INRIA", nTrees=64, cascThr=-1.5)),
                                          ("gray_face64_320x240", "gray", dict(name="FACE64", nTrees=96, cascThr=-1.0))])
def test_whole_pyramid_equals_the_frozen_reference_kernel_pyramid(oracle, name, kind, kw):
    """Every cell of every level, the hits and the mapped boxes against tests/golden/tref_pyramids.npz (the reference's compiled kernels
    on the build host), with the committed Intel tables: the device-side counterpart of tests/test_x86_tables.py's check."""
    import torch
    from acf_amd.detector import HipDetector
    fix = np.load(os.path.join(GOLD, "tref_pyramids.npz"))
    H, W, d_in, fseed, mseed = [int(v) for v in fix[name + "_meta"]]
    model = synth.make_model(seed=mseed, **kw)
    frame = synth.make_frame(fseed, H, W, kind)
    det = HipDetector(model, H, W, d_in, max_batch=1, max_hits=1 << 15)
    det.set_x86_tables(*oracle.x86_fixture())
    det.set_option("arith", 1)
    det.run(torch.from_numpy(frame[None]).cuda())
    assert np.array_equal(bits(det.read_pyramid(0)), bits(fix[name + "_pyramid_ref"]))
    d, h = det.detections(0)
    assert h.tobytes() == fix[name + "_hits_ref"].tobytes() and d.tobytes() == fix[name + "_det_ref"].tobytes() and len(h) > 0
    det.close()


@pytest.mark.parametrize("arith", [0, 1])
def test_extreme_but_finite_planes_in_both_tiers(oracle, arith):
    """Flat regions (M2 = 0: rsqrt = inf, clamped to 1e10), denormal-sized and very large gradients (squares still finite) through the
    FAST kernel forms (k_smooth_grad / _tri, k_triy_chns) of both arithmetic tiers against the oracle's tier."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 256, 512
    model = synth.make_model(seed=3, name="TINY", nTrees=64, cascThr=-2.0)
    base = synth.make_frame(21, H, W, "luv")
    frames = np.stack([base * s for s in (1.0, 1e-30, 1e-12, 1e12)]).astype(np.float32)
    frames[0, :, 100:200, :] = 0.0
    frames[0, :, :, 50:60] = 0.25           # constant: zero gradient
    frames[3, :, 300:, :] *= np.float32(1e3)
    plan = oracle.Plan(model, H, W, 3)
    oracle.set_x86_tables(*oracle.x86_fixture())
    det = HipDetector(model, H, W, 3, max_batch=4, max_hits=1 << 16)
    det.set_option("fused_grad", 2)
    det.set_option("fused_tri", 2)
    if arith:
        det.set_x86_tables(*oracle.x86_fixture())
        det.set_option("arith", 1)
    det.run(torch.from_numpy(frames).cuda())
    for f in range(4):
        oracle.set_approx(3 if arith else 0)
        try:
            want, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_approx(0)
        assert np.isfinite(want).all()
        assert np.array_equal(bits(det.read_pyramid(f)), bits(want)), (arith, f)
    det.close()
