"""Pin the oracle (oracle/acf_oracle.c) against the reference's own toolbox
kernels compiled unmodified (oracle/_ref/libacfref.so, see oracle/Makefile).

convTri1 / convTri / grad2 / gradHist involve only IEEE add/sub/mul, so the
restatement must agree BIT FOR BIT.  gradMag and gradMagNorm use
_mm_rsqrt_ps/_mm_rcp_ps in the reference (toolbox/sse.hpp:185-192; relative
error <= 1.5*2^-12 each, Intel SDM); the oracle uses exact 1/sqrt, 1/x there,
so those are checked within that bound.
"""
import ctypes as C

import numpy as np
import pytest

from acf_amd import synth

RCP_EPS = 1.5 * 2.0 ** -12

SIZES = [(64, 48), (63, 50), (48, 64), (37, 41), (120, 160), (270, 480)]


def rnd(seed, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * synth.uniform(seed, n, 3)).astype(np.float32).reshape(shape)


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("d", [1, 3])
def test_conv_tri1_bit_exact(oracle, refk, h, w, d):
    a = oracle.aligned_copy(rnd(h * 1000 + w + d, (d, w, h)))
    out_r = oracle.aligned((d, w, h))
    out_o = oracle.aligned((d, w, h))
    refk.ref_convTri1(oracle.F(a), oracle.F(out_r), h, w, d, 2.0, 1)
    assert oracle.lib().acfo_conv_tri1(oracle.F(a), oracle.F(out_o), h, w, d, 2.0, 1) == 0
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))


@pytest.mark.parametrize("h,w", SIZES)
def test_conv_tri1_aliased_bit_exact(oracle, refk, h, w):
    """The pyramid calls convTri1 with O == I (chnsCompute.cpp:239): the x pass
    then reads already-filtered columns.  The oracle must reproduce exactly that."""
    src = rnd(h * 77 + w, (3, w, h))
    a_r = oracle.aligned_copy(src)
    a_o = oracle.aligned_copy(src)
    refk.ref_convTri1(oracle.F(a_r), oracle.F(a_r), h, w, 3, 2.0, 1)
    assert oracle.lib().acfo_conv_tri1(oracle.F(a_o), oracle.F(a_o), h, w, 3, 2.0, 1) == 0
    assert np.array_equal(a_r.view(np.uint32), a_o.view(np.uint32))
    # and it really differs from the non-aliased filter (SURVEY.md H2 probe: 0.080 max-abs on U[0,1])
    out = oracle.aligned((3, w, h))
    refk.ref_convTri1(oracle.F(oracle.aligned_copy(src)), oracle.F(out), h, w, 3, 2.0, 1)
    assert np.abs(out - a_r).max() > 1e-3
    assert np.array_equal(out[:, 0], a_r[:, 0])  # column 0 is identical


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("r", [5, 2, 3])
def test_conv_tri_bit_exact(oracle, refk, h, w, r):
    a = oracle.aligned_copy(rnd(h * 31 + w * 7 + r, (1, w, h)))
    out_r = oracle.aligned((1, w, h))
    out_o = oracle.aligned((1, w, h))
    refk.ref_convTri(oracle.F(a), oracle.F(out_r), h, w, 1, r, 1)
    assert oracle.lib().acfo_conv_tri(oracle.F(a), oracle.F(out_o), h, w, 1, r, 1) == 0
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))


@pytest.mark.parametrize("h,w", SIZES)
def test_grad2_bit_exact(oracle, refk, h, w):
    a = oracle.aligned_copy(rnd(h + 3 * w, (1, w, h)))
    gx_r, gy_r = oracle.aligned((w, h)), oracle.aligned((w, h))
    gx_o, gy_o = oracle.aligned((w, h)), oracle.aligned((w, h))
    refk.ref_grad2(oracle.F(a), oracle.F(gx_r), oracle.F(gy_r), h, w, 1)
    oracle.lib().acfo_grad2(oracle.F(a), oracle.F(gx_o), oracle.F(gy_o), h, w, 1)
    assert np.array_equal(gx_r.view(np.uint32), gx_o.view(np.uint32))
    assert np.array_equal(gy_r.view(np.uint32), gy_o.view(np.uint32))


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("full", [0, 1])
def test_grad_mag_within_rcp_bound(oracle, refk, h, w, full):
    a = oracle.aligned_copy(synth.make_frame(h * w, h, w, "gray"))
    M_r, O_r = oracle.aligned((w, h)), oracle.aligned((w, h))
    M_o, O_o = oracle.aligned((w, h)), oracle.aligned((w, h))
    refk.ref_gradMag(oracle.F(a), oracle.F(M_r), oracle.F(O_r), h, w, 1, full)
    assert oracle.lib().acfo_grad_mag(oracle.F(a), oracle.F(M_o), oracle.F(O_o), h, w, 1, full) == 0
    # M = rcp(min(rsqrt(M2),1e10)): two approximate ops in the reference
    rel = np.abs(M_r - M_o) / np.maximum(np.abs(M_o), 1e-20)
    assert rel.max() <= 2.2 * RCP_EPS, rel.max()
    # orientation: the table index (Gx*m*1e4) moves by <= 1e4*RCP_EPS ~ 3.7 entries;
    # acos' slope is unbounded at +-1, so compare cos(O) instead of O
    dcos = np.abs(np.cos(O_r.astype(np.float64)) - np.cos(O_o.astype(np.float64)))
    assert dcos.max() <= 1.5 * RCP_EPS + 3e-4, dcos.max()
    # where the reference's approximate m happens to give the same table index, O must be identical
    same = np.mean(O_r == O_o)
    assert same > 0.2


def test_acos_table_matches_reference_lookup(oracle, refk):
    """Drive the reference's table through gradMag with gradients chosen so that
    Gx*m*1e4 is far from integer boundaries, and compare O bit-for-bit."""
    h, w = 8, 64
    # plane = a*x + b*y  ->  Gx = a, Gy = b (interior)
    O_all_r, O_all_o = [], []
    for k in range(50):
        ang = 0.03 + 3.08 * k / 50.0
        a_, b_ = np.cos(ang), np.sin(ang)
        x = np.arange(w, dtype=np.float64)[:, None]
        y = np.arange(h, dtype=np.float64)[None, :]
        pl = oracle.aligned_copy(((a_ * x + b_ * y) * 0.01).astype(np.float32))
        M_r, O_r = oracle.aligned((w, h)), oracle.aligned((w, h))
        M_o, O_o = oracle.aligned((w, h)), oracle.aligned((w, h))
        refk.ref_gradMag(oracle.F(pl), oracle.F(M_r), oracle.F(O_r), h, w, 1, 0)
        oracle.lib().acfo_grad_mag(oracle.F(pl), oracle.F(M_o), oracle.F(O_o), h, w, 1, 0)
        O_all_r.append(O_r[2:-2, 2:-2].copy())
        O_all_o.append(O_o[2:-2, 2:-2].copy())
    O_r, O_o = np.concatenate(O_all_r), np.concatenate(O_all_o)
    assert np.abs(O_r - O_o).max() < 2e-3  # neighbouring table entries at most
    assert np.mean(O_r == O_o) > 0.15      # and identical entries where the index coincides


@pytest.mark.parametrize("h,w", SIZES)
def test_grad_mag_norm_within_rcp_bound(oracle, refk, h, w):
    M0 = rnd(h * w + 1, (w, h), 0.0, 0.5)
    S = oracle.aligned_copy(rnd(h * w + 2, (w, h), 0.0, 0.3))
    M_r, M_o = oracle.aligned_copy(M0), oracle.aligned_copy(M0)
    refk.ref_gradMagNorm(oracle.F(M_r), oracle.F(S), h, w, 0.005)
    oracle.lib().acfo_grad_mag_norm(oracle.F(M_o), oracle.F(S), h, w, 0.005)
    rel = np.abs(M_r - M_o) / np.maximum(np.abs(M_o), 1e-20)
    assert rel.max() <= 1.1 * RCP_EPS
    n = h * w
    if n % 4:  # scalar tail divides exactly in both
        assert np.array_equal(M_r.ravel()[n - n % 4:].view(np.uint32), M_o.ravel()[n - n % 4:].view(np.uint32))


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("full", [0, 1])
@pytest.mark.parametrize("softBin", [0, 2, -2])
def test_grad_hist_bit_exact(oracle, refk, h, w, full, softBin):
    """Same M,O in -> identical histograms (accumulation order x-major, y-minor, O0 then O1), in both even-softBin branches of
    gradHist (gradientMex.cpp:391-509): orientation interpolated (>= 0), nearest bin (< 0)."""
    M = oracle.aligned_copy(rnd(h * w + 5, (w, h), 0.0, 0.6))
    hi = 2 * np.pi if full else np.pi
    O = oracle.aligned_copy(rnd(h * w + 6, (w, h), 0.0, float(hi) - 1e-6))
    hb, wb = h // 4, w // 4
    H_r, H_o = oracle.aligned((6, wb, hb)), oracle.aligned((6, wb, hb))
    refk.ref_gradHist(oracle.F(M), oracle.F(O), oracle.F(H_r), h, w, 4, 6, softBin, full)
    assert oracle.lib().acfo_grad_hist(oracle.F(M), oracle.F(O), oracle.F(H_o), h, w, 4, 6, softBin, full) == 0
    assert np.array_equal(H_r.view(np.uint32), H_o.view(np.uint32))
    assert H_r.sum() > 0


def test_grad_hist_odd_soft_bin_is_refused(oracle):
    """Odd softBin is the trilinear form with its 8/7 boundary normalisation (gradientMex.cpp:511-662): refused by the restatement
    (and by the library), not approximated."""
    h, w = 24, 20
    M = oracle.aligned_copy(rnd(991, (w, h), 0.0, 0.6))
    O = oracle.aligned_copy(rnd(992, (w, h), 0.0, float(np.pi) - 1e-6))
    for bin_, softBin in ((4, 1), (4, -1), (1, 1)):
        H_o = oracle.aligned((6, w // bin_, h // bin_))
        assert oracle.lib().acfo_grad_hist(oracle.F(M), oracle.F(O), oracle.F(H_o), h, w, bin_, 6, softBin, 0) != 0


def test_chain_with_reference_kernels(oracle, refk):
    """chnsCompute's gradient chain with every stage fed from the REFERENCE's previous
    stage: convTri1 (aliased) -> gradMag -> convTri(5) -> gradMagNorm -> gradHist.
    Bit-exact stages stay bit-exact; approximate stages stay inside the rcp bound."""
    h, w = 120, 160
    L = synth.make_frame(9, h, w, "gray")[0]
    a_r, a_o = oracle.aligned_copy(L), oracle.aligned_copy(L)
    refk.ref_convTri1(oracle.F(a_r), oracle.F(a_r), h, w, 1, 2.0, 1)
    oracle.lib().acfo_conv_tri1(oracle.F(a_o), oracle.F(a_o), h, w, 1, 2.0, 1)
    assert np.array_equal(a_r, a_o)
    M_r, O_r = oracle.aligned((w, h)), oracle.aligned((w, h))
    refk.ref_gradMag(oracle.F(a_r), oracle.F(M_r), oracle.F(O_r), h, w, 1, 0)
    S_r, S_o = oracle.aligned((w, h)), oracle.aligned((w, h))
    refk.ref_convTri(oracle.F(M_r), oracle.F(S_r), h, w, 1, 5, 1)
    oracle.lib().acfo_conv_tri(oracle.F(M_r), oracle.F(S_o), h, w, 1, 5, 1)
    assert np.array_equal(S_r.view(np.uint32), S_o.view(np.uint32))
    Mn_r, Mn_o = oracle.aligned_copy(M_r), oracle.aligned_copy(M_r)
    refk.ref_gradMagNorm(oracle.F(Mn_r), oracle.F(S_r), h, w, 0.005)
    oracle.lib().acfo_grad_mag_norm(oracle.F(Mn_o), oracle.F(S_r), h, w, 0.005)
    assert (np.abs(Mn_r - Mn_o) / np.maximum(Mn_o, 1e-20)).max() <= 1.1 * RCP_EPS
    H_r, H_o = oracle.aligned((6, w // 4, h // 4)), oracle.aligned((6, w // 4, h // 4))
    refk.ref_gradHist(oracle.F(Mn_r), oracle.F(O_r), oracle.F(H_r), h, w, 4, 6, 0, 0)
    oracle.lib().acfo_grad_hist(oracle.F(Mn_r), oracle.F(O_r), oracle.F(H_o), h, w, 4, 6, 0, 0)
    assert np.array_equal(H_r.view(np.uint32), H_o.view(np.uint32))


# ---- imResample and rgbConvert: the reference's own bodies, spliced by line range (oracle/Makefile, oracle/ref_api_*.cpp)

@pytest.fixture(scope="module")
def refk2(refk):
    if not hasattr(refk, "ref_resample"):
        pytest.skip("oracle/_ref/libacfref.so predates the resample / rgbConvert pins")
    return refk


# (ha, wa) -> (hb, wb): the geometry classes of resample (imResampleMex.cpp:145-157,198-373) — exact /2 /3 /4 per axis, generic
# down-sampling with 2..4 taps and with more than 4 taps per output (scatter form), up-sampling, identity, mixed axes, and
# the pyramid's own sizes (1080p real scales, approximated channel levels).
RESAMPLE_GEOMS = [
    ((64, 48), (32, 24)), ((66, 48), (22, 16)), ((64, 48), (16, 12)), ((64, 48), (32, 16)),
    ((64, 48), (57, 43)), ((64, 48), (40, 30)), ((63, 50), (31, 27)), ((120, 160), (13, 17)), ((97, 131), (9, 23)),
    ((32, 24), (64, 48)), ((31, 27), (63, 50)), ((30, 40), (37, 41)), ((37, 41), (37, 41)), ((64, 48), (32, 60)), ((48, 64), (60, 32)),
    ((1080, 1920), (540, 960)), ((540, 960), (272, 484)), ((270, 480), (248, 440)), ((270, 480), (136, 240)), ((68, 121), (62, 111)),
    ((135, 240), (150, 264)), ((8, 8), (4, 4)), ((5, 7), (4, 4)), ((4, 4), (9, 11)),
]


@pytest.mark.parametrize("a,b", RESAMPLE_GEOMS)
@pytest.mark.parametrize("r", [1.0, 0.8705506, 1.3195079])
def test_resample_bit_exact(oracle, refk2, a, b, r):
    (ha, wa), (hb, wb) = a, b
    d = 3 if ha * wa < 300000 else 1
    src = oracle.aligned_copy(rnd(ha * 13 + wa * 7 + hb, (d, wa, ha)))
    out_r, out_o = oracle.aligned((d, wb, hb)), oracle.aligned((d, wb, hb))
    out_r[:] = -7.0  # resample writes every output (its scatter form zeroes first): poison must disappear
    out_o[:] = -7.0
    refk2.ref_resample(oracle.F(src), oracle.F(out_r), ha, hb, wa, wb, d, r)
    assert oracle.lib().acfo_resample(oracle.F(src), oracle.F(out_o), ha, hb, wa, wb, d, r) == 0
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32)), np.abs(out_r - out_o).max()
    assert out_r.min() >= 0.0


def test_resample_chain_of_the_pyramid_bit_exact(oracle, refk2):
    """The image chain of cfg 2 at quarter size: 270x480 -> 136x240 (exact half) -> 68x121-ish generic, each fed from the
    REFERENCE's previous output."""
    src = oracle.aligned_copy(synth.make_frame(3, 272, 480, "luv"))
    cur_r, cur_o, (h, w) = src, src, (272, 480)
    for hb, wb in ((136, 240), (68, 124), (36, 60)):
        nr, no = oracle.aligned((3, wb, hb)), oracle.aligned((3, wb, hb))
        refk2.ref_resample(oracle.F(cur_r), oracle.F(nr), h, hb, w, wb, 3, 1.0)
        oracle.lib().acfo_resample(oracle.F(cur_o), oracle.F(no), h, hb, w, wb, 3, 1.0)
        assert np.array_equal(nr.view(np.uint32), no.view(np.uint32))
        cur_r, cur_o, (h, w) = nr, no, (hb, wb)


@pytest.mark.parametrize("h,w", SIZES + [(1080, 1920)])
def test_rgb2luv_vector_body_within_rcp_bound(oracle, refk2, h, w):
    """n % 4 == 0 and aligned planes: the reference takes rgb2luv_sse (rgbConvertMex.cpp:88-190), whose only approximate
    operation is one _mm_rcp_ps (:161).  L involves no rcp: bit-exact.  U, V: l * (c * x * zz - 13 * un) - minu with
    zz = rcp(..) -> the difference is bounded by RCP_EPS * l * |c * x * zz| <= RCP_EPS * |U + minu + 13 * un * l| (+ rounding)."""
    n = h * w
    if n % 4:
        pytest.skip("vector body needs n % 4 == 0")
    src = oracle.aligned_copy(synth.make_frame(h + w, h, w, "rgb"))
    out_r, out_o = oracle.aligned((3, w, h)), oracle.aligned((3, w, h))
    assert refk2.ref_rgbConvert(oracle.F(src), oracle.F(out_r), n, 3, 2, 1.0) == 0
    oracle.lib().acfo_rgb2luv(oracle.F(src), oracle.F(out_o), n)
    assert np.array_equal(out_r[0].view(np.uint32), out_o[0].view(np.uint32))
    l = out_o[0].astype(np.float64)
    for k, c in ((1, 13 * 0.197833), (2, 13 * 0.468331)):
        minc = (-88.0 if k == 1 else -134.0) / 270.0
        mag = np.abs(out_o[k].astype(np.float64) + minc + c * l)   # = l * (52 x zz) resp. l * (117 y zz)
        err = np.abs(out_r[k].astype(np.float64) - out_o[k].astype(np.float64))
        assert (err <= 1.05 * RCP_EPS * mag + 4e-7).all(), (k, err.max())
        assert err.max() > 0  # and it IS the approximate path (the bound is not vacuous)


@pytest.mark.parametrize("n", [1, 3, 17, 250, 257, 4093])
def test_rgb2luv_scalar_body_bit_exact(oracle, refk2, n):
    """n % 4 != 0: the reference's scalar rgb2luv (rgbConvertMex.cpp:62-84, true division) — the restatement's second
    branch must agree bit for bit, through rgbConvert's own dispatch and called directly."""
    assert n % 4
    src = oracle.aligned_copy(rnd(900 + n, (3, n)))
    out_r, out_d, out_o = oracle.aligned((3, n)), oracle.aligned((3, n)), oracle.aligned((3, n))
    assert refk2.ref_rgbConvert(oracle.F(src), oracle.F(out_r), n, 3, 2, 1.0) == 0
    refk2.ref_rgb2luv_scalar(oracle.F(src), oracle.F(out_d), n, 1.0)
    oracle.lib().acfo_rgb2luv(oracle.F(src), oracle.F(out_o), n)
    assert np.array_equal(out_r.view(np.uint32), out_d.view(np.uint32))
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))


def test_luv_table_bit_exact_through_the_reference(oracle, refk2):
    """rgb2luv_setup's lTable (:39-58) read back through the scalar body: grey pixels r = g = b = v give y = v * (mr1 + mg1 + mb1),
    L = lTable[(int)(y * 1024)]; sweeping v hits every table entry the conversion can reach."""
    n = 4099
    v = (np.arange(n, dtype=np.float64) / (n - 1)).astype(np.float32)
    src = oracle.aligned_copy(np.stack([v, v, v]))
    out_r, out_o = oracle.aligned((3, n)), oracle.aligned((3, n))
    refk2.ref_rgb2luv_scalar(oracle.F(src), oracle.F(out_r), n, 1.0)
    oracle.lib().acfo_rgb2luv(oracle.F(src), oracle.F(out_o), n)
    assert np.array_equal(out_r[0].view(np.uint32), out_o[0].view(np.uint32))
    assert len(np.unique(out_r[0])) > 1000


@pytest.mark.parametrize("n", [16, 250, 4099])
def test_rgb2hsv_bit_exact(oracle, refk2, n):
    """rgb2hsv (rgbConvertMex.cpp:194-238) through the reference's own rgbConvert(flag 3): every branch — grey pixels, each
    channel the maximum, ties between channels, h wrapping at 6 — bit for bit."""
    src = oracle.aligned_copy(rnd(79 + n, (3, n)))
    src[:, 0:4] = np.float32(0.25)            # r == g == b
    src[0, 4:8] = src[1, 4:8]                 # r == g
    src[1, 8:12] = src[2, 8:12]               # g == b
    src[0, 12:16] = np.maximum(src[1, 12:16], src[2, 12:16])  # r == max of the others
    out_r, out_o = oracle.aligned((3, n)), oracle.aligned((3, n))
    assert refk2.ref_rgbConvert(oracle.F(src), oracle.F(out_r), n, 3, 3, 1.0) == 0
    oracle.lib().acfo_rgb2hsv(oracle.F(src), oracle.F(out_o), n)
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))
    assert len(np.unique(out_r[0])) > 3


@pytest.mark.parametrize("n", [16, 250, 4096])
def test_rgb2gray_bit_exact(oracle, refk2, n):
    src = oracle.aligned_copy(rnd(77 + n, (3, n)))
    out_r, out_o = oracle.aligned((n,)), oracle.aligned((n,))
    assert refk2.ref_rgbConvert(oracle.F(src), oracle.F(out_r), n, 3, 0, 1.0) == 0
    oracle.lib().acfo_rgb2gray(oracle.F(src), oracle.F(out_o), n)
    assert np.array_equal(out_r.view(np.uint32), out_o.view(np.uint32))
