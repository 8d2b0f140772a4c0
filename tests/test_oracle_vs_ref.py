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
def test_grad_hist_bit_exact(oracle, refk, h, w, full):
    """Same M,O in -> identical histograms (accumulation order x-major, y-minor, O0 then O1)."""
    M = oracle.aligned_copy(rnd(h * w + 5, (w, h), 0.0, 0.6))
    hi = 2 * np.pi if full else np.pi
    O = oracle.aligned_copy(rnd(h * w + 6, (w, h), 0.0, float(hi) - 1e-6))
    hb, wb = h // 4, w // 4
    H_r, H_o = oracle.aligned((6, wb, hb)), oracle.aligned((6, wb, hb))
    refk.ref_gradHist(oracle.F(M), oracle.F(O), oracle.F(H_r), h, w, 4, 6, 0, full)
    assert oracle.lib().acfo_grad_hist(oracle.F(M), oracle.F(O), oracle.F(H_o), h, w, 4, 6, 0, full) == 0
    assert np.array_equal(H_r.view(np.uint32), H_o.view(np.uint32))
    assert H_r.sum() > 0


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
