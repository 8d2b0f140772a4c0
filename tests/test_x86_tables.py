"""The T-ref tier as a table (oracle: acfo_set_approx(3); product: acf_hip_set_x86_tables + option "arith").

`_mm_rcp_ps` / `_mm_rsqrt_ps` at the reference's three sites (T/gradientMex.cpp:209-219,266; T/rgbConvertMex.cpp:161;
T/sse.hpp:185-192) are per-CPU 12-bit approximations.  tests/golden/make_x86_tables.py showed that on the build host each is
a function of (sign, exponent / its parity, top 12 mantissa bits) for ALL 2^32 inputs and froze the 4096 + 2 x 4096 entries in
tests/golden/x86_rcp_rsqrt.npz.  Here:

 * the fixture's structure and its record of the exhaustive check;
 * the table functions against the LIVE instructions of whatever x86 CPU runs the test (tables probed live, a sample of 2^24
   inputs plus every exponent / special-value edge): holds wherever the CPU has this structure, else the test says so and skips;
 * with live tables the oracle's gradMag / gradMagNorm / rgb2luv_sse are BIT-EXACT against the reference's own compiled
   kernels (oracle/_ref) — what tests/test_oracle_vs_ref.py can only bound in the exact tier;
 * with the FIXTURE's tables the oracle reproduces tests/golden/tref_study.npz's T-ref hits (made by the reference's compiled
   kernels on the build host) bit for bit, and ref_ops.npz's frozen gradMag / gradMagNorm / rgb2luv bytes: runs anywhere.
"""
import os

import numpy as np
import pytest

from acf_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def live_tables(oracle):
    t = oracle.x86_probe()
    if t is None:
        pytest.skip("not an SSE host")
    oracle.set_x86_tables(*t)
    b = oracle.x86_verify(0, 1 << 24, 251)  # 2^24 inputs spread over the whole range
    b2 = oracle.x86_verify(0x3f000000, 1 << 20, 1)
    if b[0] or b[1] or b2[0] or b2[1]:
        pytest.skip("this CPU's rcpps / rsqrtps are not functions of the top mantissa bits: %r %r" % (b, b2))
    return t


def test_fixture_structure(oracle):
    z = np.load(oracle.X86_FIXTURE)
    rcp, rsq = z["rcp"], z["rsqrt"]
    assert rcp.shape == (4096,) and rsq.shape == (8192,) and rcp.dtype == np.uint32
    assert list(z["checked"]) == [1 << 32, 0, 0]          # every input, no mismatch, on the CPU named in `cpu`
    assert len(str(z["cpu"])) > 4
    assert not (rcp & 0x7ff).any() and not (rsq & 0x7ff).any()  # 12 significant bits
    f = rcp.view(np.float32)
    x = 1.0 + np.arange(4096) / 4096.0
    assert (np.abs(f * (x + 1.0 / 8192) - 1.0) < 1.5 * 2.0 ** -12).all()   # within the documented bound at the cell centres
    assert (np.diff(f.astype(np.float64)) <= 0).all()
    g = rsq.view(np.float32).astype(np.float64)
    xs = np.concatenate([1.0 + (np.arange(4096) + 0.5) / 4096.0, 2.0 + 2.0 * (np.arange(4096) + 0.5) / 4096.0])
    assert (np.abs(g * np.sqrt(xs) - 1.0) < 1.5 * 2.0 ** -12).all()
    assert (np.diff(g[:4096]) <= 0).all() and (np.diff(g[4096:]) <= 0).all()


def test_second_fixture_from_the_gpu_boxes_host(oracle):
    """The same probe + exhaustive check run on a GPU box's host (AMD EPYC 9575F, tests/golden/make_x86_tables.py <path> there): another
    vendor's instructions are table functions of the same 12 mantissa bits with other entries — which is why the tier takes tables and
    the host class probes the CPU it runs on (tests/test_gpu_arith.py compares the HIP path with the reference's compiled kernels LIVE
    on whatever host it runs on)."""
    a = np.load(oracle.X86_FIXTURE)
    b = np.load(os.path.join(HERE, "golden", "x86_rcp_rsqrt_amd_epyc_9575f.npz"))
    assert list(b["checked"]) == [1 << 32, 0, 0] and "EPYC" in str(b["cpu"])
    assert b["rcp"].shape == (4096,) and b["rsqrt"].shape == (8192,)
    assert not (b["rcp"] & 0x7ff).any() and not (b["rsqrt"] & 0x7ff).any()
    assert (a["rcp"] != b["rcp"]).sum() > 1000 and (a["rsqrt"] != b["rsqrt"]).sum() > 1000      # different CPUs, different bits
    x = 1.0 + (np.arange(4096) + 0.5) / 4096.0
    for t in (a, b):                                                                             # ... both inside the documented bound
        assert (np.abs(t["rcp"].view(np.float32).astype(np.float64) * x - 1.0) < 1.5 * 2.0 ** -12).all()


def test_table_functions_special_values(oracle):
    oracle.set_x86_tables(*oracle.x86_fixture())
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    r = oracle.x86_rcp(np.asarray([0.0, -0.0, inf, -inf, 1e-45, 1.0, 2.0 ** 127, 3e38], np.float32))
    assert r[0] == inf and r[1] == -inf and r[2] == 0 and r[3] == 0 and np.signbit(r[3]) and r[4] == inf
    assert abs(r[5] - 1.0) < 4e-4 and r[7] == 0.0
    q = oracle.x86_rsqrt(np.asarray([0.0, -0.0, inf, -1.0, 4.0, 1e-45], np.float32))
    assert q[0] == inf and q[1] == -inf and q[2] == 0 and np.isnan(q[3]) and abs(q[4] - 0.5) < 2e-4 and q[5] == inf
    assert np.isnan(oracle.x86_rcp(np.asarray([nan], np.float32))[0])


def test_table_functions_equal_the_live_instructions(oracle, live_tables):
    # (the fixture itself did this for all 2^32 inputs in make_x86_tables.py; here a spread sample on the CPU at hand)
    assert oracle.x86_verify(0, 1 << 24, 255) == (0, 0)
    assert oracle.x86_verify(0x7f000000, 1 << 24, 1) == (0, 0)   # the top exponents: underflow of rcp, inf, NaN
    assert oracle.x86_verify(0x00000000, 1 << 24, 1) == (0, 0)   # zero, subnormals, the smallest normals
    assert oracle.x86_verify(0xff000000, 1 << 24, 1) == (0, 0)


def test_fixture_is_this_hosts_when_the_cpu_matches(oracle):
    t = oracle.x86_probe()
    if t is None:
        pytest.skip("not an SSE host")
    z = np.load(oracle.X86_FIXTURE)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    if model not in str(z["cpu"]):
        pytest.skip("fixture from another CPU (%s), this is %s" % (z["cpu"], model))
    assert np.array_equal(t[0], z["rcp"]) and np.array_equal(t[1], z["rsqrt"])


@pytest.fixture(scope="module")
def refk2(refk):
    if not hasattr(refk, "ref_rgbConvert"):
        pytest.skip("oracle/_ref/libacfref.so predates the rgbConvert pin")
    return refk


SIZES = [(64, 48), (63, 50), (48, 64), (37, 41), (120, 160), (270, 480)]


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("full", [0, 1])
def test_grad_mag_table_tier_bit_exact_vs_reference(oracle, refk, live_tables, h, w, full):
    a = oracle.aligned_copy(synth.make_frame(h * w, h, w, "gray"))
    M_r, O_r = oracle.aligned((w, h)), oracle.aligned((w, h))
    M_o, O_o = oracle.aligned((w, h)), oracle.aligned((w, h))
    refk.ref_gradMag(oracle.F(a), oracle.F(M_r), oracle.F(O_r), h, w, 1, full)
    oracle.set_approx(3)
    try:
        assert oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(M_o), oracle.F(O_o), h, w, 1, full) == 0
    finally:
        oracle.set_approx(0)
    assert np.array_equal(M_r.view(np.uint32), M_o.view(np.uint32))
    assert np.array_equal(O_r.view(np.uint32), O_o.view(np.uint32))


def test_grad_mag_table_tier_flat_and_extreme_planes(oracle, refk, live_tables):
    """M2 = 0 (rsqrt = inf, clamped to 1e10, M = rcp(1e10)), tiny and huge gradients."""
    h, w = 16, 12
    for scale in (0.0, 1e-30, 1e-12, 1.0, 1e15):
        a = oracle.aligned_copy((synth.make_frame(7, h, w, "gray") * scale).astype(np.float32))
        a[:, :4] = 0
        M_r, O_r = oracle.aligned((w, h)), oracle.aligned((w, h))
        M_o, O_o = oracle.aligned((w, h)), oracle.aligned((w, h))
        refk.ref_gradMag(oracle.F(a), oracle.F(M_r), oracle.F(O_r), h, w, 1, 0)
        oracle.set_approx(3)
        try:
            oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(M_o), oracle.F(O_o), h, w, 1, 0)
        finally:
            oracle.set_approx(0)
        assert np.array_equal(M_r.view(np.uint32), M_o.view(np.uint32)), scale
        assert np.array_equal(O_r.view(np.uint32), O_o.view(np.uint32)), scale


@pytest.mark.parametrize("h,w", SIZES)
def test_grad_mag_norm_table_tier_bit_exact_vs_reference(oracle, refk, live_tables, h, w):
    rnd = lambda seed, lo, hi: (lo + (hi - lo) * synth.uniform(seed, h * w, 3)).astype(np.float32).reshape(w, h)
    M0 = rnd(h * w + 1, 0.0, 0.5)
    S = oracle.aligned_copy(rnd(h * w + 2, 0.0, 0.3))
    M_r, M_o = oracle.aligned_copy(M0), oracle.aligned_copy(M0)
    refk.ref_gradMagNorm(oracle.F(M_r), oracle.F(S), h, w, 0.005)
    oracle.set_approx(3)
    try:
        oracle.lib().acfo_grad_mag_norm(oracle.F(M_o), oracle.F(S), h, w, 0.005)
    finally:
        oracle.set_approx(0)
    assert np.array_equal(M_r.view(np.uint32), M_o.view(np.uint32))   # vector body AND the n % 4 scalar tail


@pytest.mark.parametrize("h,w", SIZES + [(1080, 1920)])
def test_rgb2luv_table_tier_bit_exact_vs_reference(oracle, refk2, live_tables, h, w):
    n = h * w
    src = oracle.aligned_copy(synth.make_frame(h + w, h, w, "rgb"))
    out_r, out_o = oracle.aligned((3, w, h)), oracle.aligned((3, w, h))
    assert refk2.ref_rgbConvert(oracle.F(src), oracle.F(out_r), n, 3, 2, 1.0) == 0
    oracle.set_approx(3)
    try:
        oracle.lib().acfo_rgb2luv(oracle.F(src), oracle.F(out_o), n)
    finally:
        oracle.set_approx(0)
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))   # (n % 4 != 0: the scalar body, exact in every tier)


@pytest.mark.parametrize("shape", [("FACE80", 540, 960, "luv", 3), ("INRIA", 240, 320, "rgb", 3), ("FACE64", 240, 320, "gray", 1)])
def test_table_tier_pyramid_equals_reference_kernel_pyramid(oracle, refk, live_tables, shape):
    """The whole pyramid: restated orchestration + table tier == restated orchestration + the reference's compiled kernels."""
    name, H, W, kind, d = shape
    model = synth.make_model(seed=3, name=name, nTrees=64)
    frame = synth.make_frame(5, H, W, kind)
    plan = oracle.Plan(model, H, W, d)
    oracle.set_tref(True)
    try:
        ref, _, _ = oracle.chns_pyramid(plan, frame)
    finally:
        oracle.set_tref(False)
    oracle.set_approx(3)
    try:
        got, _, _ = oracle.chns_pyramid(plan, frame)
    finally:
        oracle.set_approx(0)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


TREF_CFG = {
    "cfg1_vga_gray_face64": ("gray", "FACE64"),
    "cfg2_1080p_luv_face80": ("luv", "FACE80"),
    "cfg4_vga_rgb_inria": ("rgb", "INRIA"),
}


@pytest.mark.parametrize("cfg", list(TREF_CFG))
def test_fixture_tables_reproduce_the_reference_kernels_hits(oracle, cfg):
    """No reference needed: the committed tables + the oracle reproduce, bit for bit, the hits the reference's own compiled
    kernels gave on the build host (tests/golden/tref_study.npz `*_hits_ref`), at BASELINE.json's full sizes."""
    fix = np.load(os.path.join(HERE, "golden", "tref_study.npz"))
    H, W, d_in, nframes, seed0, mseed = [int(v) for v in fix[cfg + "_meta"]]
    kind, preset = TREF_CFG[cfg]
    model = synth.make_model(seed=mseed, name=preset)
    plan = oracle.Plan(model, H, W, d_in)
    oracle.set_x86_tables(*oracle.x86_fixture())
    for f in range(2 if H > 600 else 4):
        frame = synth.make_frame(seed0 + f, H, W, kind)
        oracle.set_approx(3)
        try:
            pyr, _, _ = oracle.chns_pyramid(plan, frame)
        finally:
            oracle.set_approx(0)
        _, hits = oracle.detect(plan, pyr)
        want = fix["%s_f%d_hits_ref" % (cfg, f)]
        assert hits.tobytes() == want.tobytes(), (cfg, f, len(hits), len(want))


def test_fixture_tables_reproduce_the_frozen_reference_bytes(oracle):
    """ref_ops.npz / ref_resample_luv.npz hold the reference's compiled gradMag, gradMagNorm and rgbConvert outputs from the build
    host: with the fixture's tables the oracle equals them bit for bit (the GPU box repeats this on the device: test_gpu_arith.py)."""
    ops = np.load(os.path.join(HERE, "golden", "ref_ops.npz"))
    rsl = np.load(os.path.join(HERE, "golden", "ref_resample_luv.npz"))
    oracle.set_x86_tables(*oracle.x86_fixture())
    oracle.set_approx(3)
    try:
        for k, (h, w) in enumerate(ops["sizes"]):
            h, w = int(h), int(w)
            a = oracle.aligned_copy(ops["in%d" % k][0])
            M, O = oracle.aligned((w, h)), oracle.aligned((w, h))
            assert oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(M), oracle.F(O), h, w, 1, 0) == 0
            assert np.array_equal(M.view(np.uint32), ops["M_%d" % k].view(np.uint32)), k
            assert np.array_equal(O.view(np.uint32), ops["O_%d" % k].view(np.uint32)), k
            if "Mn_%d" % k in ops.files:
                Mn = oracle.aligned_copy(ops["M_%d" % k])
                S = oracle.aligned_copy(ops["S_%d" % k])
                oracle.lib().acfo_grad_mag_norm(oracle.F(Mn), oracle.F(S), h, w, 0.005)
                assert np.array_equal(Mn.view(np.uint32), ops["Mn_%d" % k].view(np.uint32)), k
        for k, (h, w) in enumerate(rsl["luv_sizes"]):
            h, w = int(h), int(w)
            a = oracle.aligned_copy(synth.make_frame(600 + k, h, w, "rgb"))
            luv = oracle.aligned((3, w, h))
            oracle.lib().acfo_rgb2luv(oracle.F(a), oracle.F(luv), h * w)
            assert np.array_equal(luv.view(np.uint32), rsl["luv_out%d" % k].view(np.uint32)), k
    finally:
        oracle.set_approx(0)


@pytest.mark.parametrize("name,kind,kw", [("luv_tiny_160x120", "luv", dict(name="TINY", nTrees=64, cascThr=-3.0)),
                                          ("rgb_inria_160x120", "rgb", dict(name="INRIA", nTrees=64, cascThr=-1.5)),
                                          ("gray_face64_320x240", "gray", dict(name="FACE64", nTrees=96, cascThr=-1.0))])
def test_fixture_tables_reproduce_whole_reference_kernel_pyramids(oracle, name, kind, kw):
    """tests/golden/tref_pyramids.npz (make_tref.py --small-only, build host): every cell of every level of the fused pyramid as the
    reference's own compiled kernels give it, and the hits on it.  The oracle's table tier with the committed Intel tables must equal
    it bit for bit on any box (tests/test_gpu_arith.py holds the device against the same bytes)."""
    fix = np.load(os.path.join(HERE, "golden", "tref_pyramids.npz"))
    H, W, d_in, fseed, mseed = [int(v) for v in fix[name + "_meta"]]
    model = synth.make_model(seed=mseed, **kw)
    frame = synth.make_frame(fseed, H, W, kind)
    plan = oracle.Plan(model, H, W, d_in)
    oracle.set_x86_tables(*oracle.x86_fixture())
    oracle.set_approx(3)
    try:
        pyr, _, _ = oracle.chns_pyramid(plan, frame)
    finally:
        oracle.set_approx(0)
    det, hits = oracle.detect(plan, pyr)
    assert np.array_equal(pyr.view(np.uint32), fix[name + "_pyramid_ref"].view(np.uint32))
    assert hits.tobytes() == fix[name + "_hits_ref"].tobytes() and det.tobytes() == fix[name + "_det_ref"].tobytes()
    assert len(hits) > 0
