"""ctypes binding of oracle/liboracle.so and (when present) oracle/_ref/libacfref.so.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from acf_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libacfref.so")

fp = C.POINTER(C.c_float)


class Taps(C.Structure):
    _fields_ = [("image", fp), ("smoothed", fp), ("M", fp), ("O", fp), ("S", fp), ("Mnorm", fp)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


_o = None
_r = None


def lib():
    global _o
    if _o is None:
        if not os.path.exists(ORACLE_SO):
            build()
        o = C.CDLL(ORACLE_SO)
        P = C.POINTER(capi.Params)
        L = C.POINTER(capi.Level)
        dp = C.POINTER(C.c_double)
        o.acfo_get_scales.argtypes = [C.c_int] * 7 + [dp, dp, dp, C.c_int]
        o.acfo_get_scales.restype = C.c_int
        o.acfo_resample.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        o.acfo_resample.restype = C.c_int
        o.acfo_luv_table.restype = fp
        o.acfo_acos_table.restype = fp
        o.acfo_rgb2luv.argtypes = [fp, fp, C.c_int]
        o.acfo_rgb2gray.argtypes = [fp, fp, C.c_int]
        o.acfo_rgb2hsv.argtypes = [fp, fp, C.c_int]
        o.acfo_rgb2hsv.restype = None
        o.acfo_ingest_u8.argtypes = [C.c_void_p] + [C.c_int] * 7 + [fp, C.c_int]
        o.acfo_ingest_u8.restype = None
        o.acfo_conv_tri1.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        o.acfo_conv_tri1.restype = C.c_int
        o.acfo_conv_tri.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        o.acfo_conv_tri.restype = C.c_int
        o.acfo_conv_tri_dispatch.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        o.acfo_conv_tri_dispatch.restype = C.c_int
        o.acfo_grad2.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int]
        o.acfo_grad_mag.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        o.acfo_grad_mag.restype = C.c_int
        o.acfo_grad_mag_norm.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float]
        o.acfo_grad_hist.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        o.acfo_grad_hist.restype = C.c_int
        o.acfo_plan.argtypes = [P, C.c_int, C.c_int, C.c_int, L, C.c_int, C.POINTER(C.c_int)]
        o.acfo_plan.restype = C.c_int
        o.acfo_chns_compute.argtypes = [fp, C.c_int, C.c_int, C.c_int, P, fp, C.POINTER(Taps)]
        o.acfo_chns_compute.restype = C.c_int
        o.acfo_chns_compute_mo.argtypes = [fp, C.c_int, C.c_int, C.c_int, P, fp, fp]
        o.acfo_chns_compute_mo.restype = C.c_int
        o.acfo_chns_pyramid.argtypes = [fp, C.c_int, C.c_int, C.c_int, P, L, C.c_int, fp, C.POINTER(Taps), C.POINTER(fp)]
        o.acfo_chns_pyramid.restype = C.c_int
        o.acfo_nms.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                               C.c_double, C.POINTER(C.c_int32)]
        o.acfo_nms.restype = C.c_int
        o.acfo_plane_sum.argtypes = [fp, C.c_int]
        o.acfo_plane_sum.restype = C.c_double
        o.acfo_lambda.argtypes = [C.c_double] * 6
        o.acfo_lambda.restype = C.c_double
        o.acfo_lambda_levels.argtypes = [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        o.acfo_lambda_levels.restype = C.c_int
        o.acfo_last_lambdas.argtypes = [C.POINTER(C.c_double)]
        o.acfo_last_lambdas.restype = None
        o.acfo_acf_detect1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, P,
                                       C.POINTER(capi.Hit), C.c_int, C.c_int]
        o.acfo_acf_detect1.restype = C.c_int
        o.acfo_evaluate.argtypes = [fp, C.c_int, C.c_int, C.c_int, P, C.c_float]
        o.acfo_evaluate.restype = C.c_float
        o.acfo_ldcf_plan.argtypes = [P, L, C.c_int, C.c_int, L]
        o.acfo_ldcf_plan.restype = C.c_int64
        o.acfo_ldcf_pyramid.argtypes = [fp, P, L, L, C.c_int, C.c_int, fp]
        o.acfo_ldcf_pyramid.restype = C.c_int
        o.acfo_resize_dims.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        o.acfo_resize_dims.restype = None
        o.acfo_resize_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_int]
        o.acfo_resize_u8.restype = C.c_int
        o.acfo_unscale_rect.argtypes = [C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        o.acfo_unscale_rect.restype = None
        o.acfo_thrs_u8.argtypes = [fp, C.c_int, C.c_void_p]
        o.acfo_thrs_u8.restype = None
        o.acfo_mean_trees.argtypes = [fp, C.c_int, C.c_int, C.c_int, P]
        o.acfo_mean_trees.restype = C.c_double
        o.acfo_detect.argtypes = [fp, P, L, C.c_int, C.c_int, C.POINTER(capi.Detection), C.POINTER(capi.Hit), C.c_int]
        o.acfo_detect.restype = C.c_int
        _o = o
    return _o


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own toolbox kernels (oracle/_ref/libacfref.so)."""
    global _r
    if _r is None:
        r = C.CDLL(REF_SO)
        r.ref_convTri.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        r.ref_convTri1.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        r.ref_grad2.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int]
        r.ref_gradMag.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        r.ref_gradMagNorm.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float]
        r.ref_gradHist.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        for f in (r.ref_convTri, r.ref_convTri1, r.ref_grad2, r.ref_gradMag, r.ref_gradMagNorm, r.ref_gradHist):
            f.restype = None
        if hasattr(r, "ref_resample"):  # (a prebuilt _ref from before round 3 has only the six kernels above)
            r.ref_resample.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
            r.ref_resample.restype = None
            r.ref_rgbConvert.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float]
            r.ref_rgbConvert.restype = C.c_int
            r.ref_rgb2luv_scalar.argtypes = [fp, fp, C.c_int, C.c_float]
            r.ref_rgb2luv_scalar.restype = None
        _r = r
    return _r


class RefKernels(C.Structure):
    """acfo_ref_kernels (oracle/acf_oracle.c): addresses of the reference's own compiled toolbox kernels."""
    _fields_ = [(n, C.c_void_p) for n in ("convTri", "convTri1", "gradMag", "gradMagNorm", "gradHist", "resample", "rgbConvert")]


TREF_ALL = ("convTri", "convTri1", "gradMag", "gradMagNorm", "gradHist", "resample", "rgbConvert")
TREF_APPROX = ("gradMag", "gradMagNorm", "rgbConvert")  # the kernels that hold an _mm_rsqrt_ps / _mm_rcp_ps


def set_tref(on, only=TREF_ALL):
    """T-ref mode of the oracle's orchestration (build container only): chns_pyramid() on THIS thread calls the reference's
    own compiled kernels (oracle/_ref: rsqrtps / rcpps and all) instead of the T-exact restatements — all of them, or the
    ones named in `only`.  set_tref(False) restores."""
    o = lib()
    o.acfo_set_ref_kernels.argtypes = [C.c_void_p]
    o.acfo_set_ref_kernels.restype = None
    if not on:
        o.acfo_set_ref_kernels(None)
        return
    r = ref()
    k = RefKernels()
    for name, fn in (("convTri", r.ref_convTri), ("convTri1", r.ref_convTri1), ("gradMag", r.ref_gradMag),
                     ("gradMagNorm", r.ref_gradMagNorm), ("gradHist", r.ref_gradHist), ("resample", r.ref_resample),
                     ("rgbConvert", r.ref_rgbConvert)):
        if name in only:
            setattr(k, name, C.cast(fn, C.c_void_p).value)
    o.acfo_set_ref_kernels(C.byref(k))


def set_approx(mode):
    """T-approx tier of the oracle on THIS thread (tests/golden/make_tref.py): 0 exact (default), 1 / 2 the three rsqrt / rcp sites as
    12-bit approximations (rounded / truncated), two more implementations within _mm_rsqrt_ps's documented error bound; 3 the sites
    as the TABLES of one CPU's _mm_rcp_ps / _mm_rsqrt_ps (set_x86_tables first): that CPU's T-ref bits."""
    o = lib()
    o.acfo_set_approx.argtypes = [C.c_int]
    o.acfo_set_approx.restype = None
    o.acfo_set_approx(int(mode))


X86_FIXTURE = os.path.join(os.path.dirname(_HERE), "tests", "golden", "x86_rcp_rsqrt.npz")
u32p = C.POINTER(C.c_uint32)


def _x86_lib():
    o = lib()
    if not getattr(o, "_x86_bound", False):
        o.acfo_set_x86_tables.argtypes = [u32p, u32p]
        o.acfo_set_x86_tables.restype = None
        o.acfo_x86_probe.argtypes = [u32p, u32p]
        o.acfo_x86_probe.restype = C.c_int
        o.acfo_x86_verify.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        o.acfo_x86_verify.restype = C.c_int
        o.acfo_x86_digest.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        o.acfo_x86_digest.restype = None
        o.acfo_x86_rcp_bits.argtypes = [C.c_uint32]
        o.acfo_x86_rcp_bits.restype = C.c_uint32
        o.acfo_x86_rsqrt_bits.argtypes = [C.c_uint32]
        o.acfo_x86_rsqrt_bits.restype = C.c_uint32
        o._x86_bound = True
    return o


def x86_probe():
    """(rcp[4096], rsqrt[8192]) uint32 tables of the CPU THIS process runs on (_mm_rcp_ps over [1, 2), _mm_rsqrt_ps over [1, 4)),
    or None where the oracle was not built for an SSE host."""
    o = _x86_lib()
    rcp, rsq = np.zeros(4096, np.uint32), np.zeros(8192, np.uint32)
    if not o.acfo_x86_probe(rcp.ctypes.data_as(u32p), rsq.ctypes.data_as(u32p)):
        return None
    return rcp, rsq


def x86_fixture():
    """The committed tables (tests/golden/x86_rcp_rsqrt.npz, made by tests/golden/make_x86_tables.py on the build host): the
    arithmetic tests/golden/tref_study.npz's T-ref hits and ref_ops.npz's gradMag / gradMagNorm / rgb2luv_sse bytes were made with."""
    z = np.load(X86_FIXTURE)
    return np.ascontiguousarray(z["rcp"], np.uint32), np.ascontiguousarray(z["rsqrt"], np.uint32)


def set_x86_tables(rcp, rsq):
    """Install the tables acfo_set_approx(3) evaluates (process-wide)."""
    o = _x86_lib()
    rcp = np.ascontiguousarray(rcp, np.uint32)
    rsq = np.ascontiguousarray(rsq, np.uint32)
    assert rcp.shape == (4096,) and rsq.shape == (8192,)
    o.acfo_set_x86_tables(rcp.ctypes.data_as(u32p), rsq.ctypes.data_as(u32p))


def x86_verify(first, count, stride=1):
    """(#rcp mismatches, #rsqrt mismatches) of the installed tables against the live instructions over first + i * stride."""
    o = _x86_lib()
    bad = (C.c_uint64 * 2)()
    if not o.acfo_x86_verify(C.c_uint32(first), C.c_uint64(count), C.c_uint32(stride), bad):
        return None
    return int(bad[0]), int(bad[1])


def x86_digest(first, count, stride=1):
    """Position-mixed 64-bit digests (rcp, rsqrt) of the installed tables' functions over first + i * stride (acfo_x86_digest)."""
    o = _x86_lib()
    out = (C.c_uint64 * 2)()
    o.acfo_x86_digest(C.c_uint32(first), C.c_uint64(count), C.c_uint32(stride), out)
    return int(out[0]), int(out[1])


def x86_rcp(x):
    o = _x86_lib()
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).ravel()
    return np.asarray([o.acfo_x86_rcp_bits(int(v)) for v in b], np.uint32).view(np.float32)


def x86_rsqrt(x):
    o = _x86_lib()
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).ravel()
    return np.asarray([o.acfo_x86_rsqrt_bits(int(v)) for v in b], np.uint32).view(np.float32)


def aligned(shape, dtype=np.float32, align=64):
    """numpy array whose data pointer is `align`-byte aligned (the reference's
    SSE paths require 16-byte alignment, like cv::Mat storage)."""
    n = int(np.prod(shape))
    item = np.dtype(dtype).itemsize
    raw = np.zeros(n * item + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * item].view(dtype).reshape(shape)


def aligned_copy(a):
    b = aligned(a.shape, a.dtype)
    b[...] = a
    return b


F = capi.fptr

# ---------------------------------------------------------------- high level


_PIX = {0: (3, 0, 1, 2), 1: (3, 2, 1, 0), 2: (4, 0, 1, 2), 3: (4, 2, 1, 0), 4: (1, 0, 0, 0)}  # ACF_HIP_PIX_* -> cpp, ro, go, bo


def ingest_u8(img, pix, row_stride=0):
    """Packed upright uint8 image [H][stride bytes] or [H][W][cpp] -> planar transposed f32 [d][W][H] (acfo_ingest_u8)."""
    cpp, ro, go, bo = _PIX[pix]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H = img.shape[0]
    stride = row_stride or int(np.prod(img.shape[1:]))
    W = img.shape[1] if img.ndim == 3 or cpp == 1 and not row_stride else None
    if W is None:
        raise ValueError("pass [H][W][cpp] images, or [H][W] for GRAY")
    n_out = 1 if cpp == 1 else 3
    out = aligned((n_out, W, H))
    lib().acfo_ingest_u8(img.ctypes.data, H, W, cpp, ro, go, bo, stride, F(out), n_out)
    return out


class Plan:
    def __init__(self, model, H, W, d_in):
        self.model = model
        self.params, self._keep = capi.make_params(model)
        self.H, self.W, self.d_in = H, W, d_in
        lv = (capi.Level * 256)()
        nC = C.c_int(0)
        n = lib().acfo_plan(C.byref(self.params), H, W, d_in, lv, 256, C.byref(nC))
        assert 0 <= n <= 256
        self.levels = lv
        self.nScales = n
        self.nChns = nC.value
        self.total = sum(self.nChns * lv[i].hP * lv[i].wP for i in range(n))
        self.real = [i for i in range(n) if lv[i].isReal]

    def level_view(self, pyr, i):
        l = self.levels[i]
        return pyr[l.offset:l.offset + self.nChns * l.hP * l.wP].reshape(self.nChns, l.wP, l.hP)


def chns_pyramid(plan, frame, want_taps=False, want_chns=False):
    """Oracle chnsPyramid on one frame [d][W][H] -> (flat pyramid, taps, chns)."""
    o = lib()
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    out = np.zeros(plan.total, dtype=np.float32)
    taps_c = None
    taps = None
    lv = plan.levels
    d = 1 if plan.model["colorSpace"] == capi.CS_GRAY else 3
    if want_taps:
        taps_c = (Taps * len(plan.real))()
        taps = []
        sh = plan.model["shrink"]
        for k, i in enumerate(plan.real):
            h1, w1 = lv[i].hC * sh, lv[i].wC * sh
            t = dict(image=np.zeros((d, w1, h1), np.float32), smoothed=np.zeros((d, w1, h1), np.float32),
                     M=np.zeros((w1, h1), np.float32), O=np.zeros((w1, h1), np.float32),
                     S=np.zeros((w1, h1), np.float32), Mnorm=np.zeros((w1, h1), np.float32))
            for name in t:
                setattr(taps_c[k], name, F(t[name]))
            taps.append(t)
    chns_c = None
    chns = None
    if want_chns:
        chns = [np.zeros((plan.nChns, lv[i].wC, lv[i].hC), np.float32) for i in range(plan.nScales)]
        chns_c = (fp * plan.nScales)(*[F(a) for a in chns])
    rc = o.acfo_chns_pyramid(F(frame), plan.H, plan.W, plan.d_in, C.byref(plan.params), lv, plan.nScales, F(out),
                             taps_c, chns_c)
    if rc:
        raise RuntimeError("acfo_chns_pyramid rc=%d" % rc)
    return out, taps, chns


def chns_compute(model, frame):
    """Detector::chnsCompute as the oracle restates it, on one image [d][w][h]: crop to a multiple of shrink (chnsCompute.cpp:203-217),
    rgbConvert (:235; acfo_rgb2luv / acfo_rgb2gray / acfo_rgb2hsv, skipped for orig / rgb / isLuv), then acfo_chns_compute (convTri in
    place, gradMag, gradHist, addChn).  -> [nChns][w / shrink][h / shrink]."""
    o = lib()
    prm, keep = capi.make_params(model)
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    d, w, h = frame.shape
    sh = int(prm.shrink)
    hc, wc = h - h % sh, w - w % sh
    MO = None
    if d == 5:   # the image's own M, O planes (chnsCompute.cpp:219-226)
        MO = aligned_copy(frame[3:, :wc, :hc])
        frame = frame[:3]
        d = 3
    I = aligned_copy(frame[:, :wc, :hc])
    n = hc * wc
    cs = int(prm.colorSpace)
    if d == 3 and (cs in (capi.CS_ORIG, capi.CS_RGB) or (prm.isLuv and cs == capi.CS_LUV)):
        col = I
    elif cs == capi.CS_LUV:
        col = aligned((3, wc, hc))
        o.acfo_rgb2luv(F(I), F(col), n)
    elif cs == capi.CS_GRAY:
        col = aligned((1, wc, hc))
        src = I if d == 3 else aligned_copy(np.repeat(I, 3, axis=0))
        o.acfo_rgb2gray(F(src), F(col), n)
    elif cs == capi.CS_HSV:
        col = aligned((3, wc, hc))
        o.acfo_rgb2hsv(F(I), F(col), n)
    else:
        col = aligned_copy(np.repeat(I, 3, axis=0))
    dcol = col.shape[0]
    nC = (dcol if prm.colorEnabled else 0) + (1 if prm.gradMagEnabled else 0) + (int(prm.nOrients) if prm.gradHistEnabled else 0)
    out = aligned((nC, wc // sh, hc // sh))
    if MO is not None:
        rc = o.acfo_chns_compute_mo(F(col), hc, wc, dcol, C.byref(prm), F(out), F(MO))
    else:
        rc = o.acfo_chns_compute(F(col), hc, wc, dcol, C.byref(prm), F(out), None)
    if rc:
        raise RuntimeError("acfo_chns_compute rc=%d" % rc)
    return np.array(out)


def nms(boxes, scores, params):
    """Oracle bbNms + prune: boxes int32 [n][4], scores float64 [n], params capi.NmsParams -> indices of the survivors in order."""
    boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    keep = np.zeros(max(len(scores), 1), np.int32)
    m = lib().acfo_nms(boxes.ctypes.data_as(C.POINTER(C.c_int32)), scores.ctypes.data_as(C.POINTER(C.c_double)), len(scores), params.type,
                       params.ovrDnmUnion, params.overlap, params.thr, params.prune, params.maxCount, params.pruneRatio,
                       keep.ctypes.data_as(C.POINTER(C.c_int32)))
    return keep[:m].copy()


def last_lambdas():
    """The three lambdas the last chns_pyramid call on this thread used (the model's, or estimated from the image)."""
    out = (C.c_double * 3)()
    lib().acfo_last_lambdas(out)
    return [out[0], out[1], out[2]]


def detect(plan, pyr, cap=1 << 16):
    o = lib()
    det = np.zeros(cap, dtype=capi.DET_DTYPE)
    hits = np.zeros(cap, dtype=capi.HIT_DTYPE)
    n = o.acfo_detect(F(pyr), C.byref(plan.params), plan.levels, plan.nScales, plan.nChns,
                      det.ctypes.data_as(C.POINTER(capi.Detection)), hits.ctypes.data_as(C.POINTER(capi.Hit)), cap)
    if n > cap:
        raise RuntimeError("capacity %d < %d" % (cap, n))
    return det[:n].copy(), hits[:n].copy()


def ldcf(plan, pyr):
    """LDCF post-stage of the pyramid (acfo_ldcf_*): -> (levels, ldcf pyramid, params with shrink*2 for the cascade)."""
    n = plan.nScales
    lvL = (capi.Level * len(plan.levels))()
    k = plan.params.ldcfK
    tot = lib().acfo_ldcf_plan(C.byref(plan.params), plan.levels, n, plan.nChns, lvL)
    out = aligned((int(tot),))
    rc = lib().acfo_ldcf_pyramid(F(pyr), C.byref(plan.params), plan.levels, lvL, n, plan.nChns, F(out))
    if rc:
        raise RuntimeError("acfo_ldcf_pyramid rc=%d" % rc)
    return lvL, out, k


def detect_ldcf(plan, lvL, pyrL, cap=1 << 16):
    """Cascade + box mapping on the LDCF pyramid: shrink*2, nChns*k channels."""
    import copy
    p2 = capi.Params.from_buffer_copy(plan.params)
    p2.shrink = plan.params.shrink * 2
    det = np.zeros(cap, dtype=capi.DET_DTYPE)
    hits = np.zeros(cap, dtype=capi.HIT_DTYPE)
    n = lib().acfo_detect(F(pyrL), C.byref(p2), lvL, plan.nScales, plan.nChns * plan.params.ldcfK,
                          det.ctypes.data_as(C.POINTER(capi.Detection)), hits.ctypes.data_as(C.POINTER(capi.Hit)), cap)
    n = min(n, cap)
    return det[:n].copy(), hits[:n].copy()


def resize_dims(rows, cols, scale):
    r, c_ = C.c_int(), C.c_int()
    lib().acfo_resize_dims(rows, cols, C.c_double(scale), C.c_double(scale), C.byref(r), C.byref(c_))
    return r.value, c_.value


def resize_u8(img, scale, interp=None):
    """cv::resize(img, {}, scale, scale, INTER_AREA if scale < 1 else INTER_LINEAR) as restated by acfo_resize_u8 (the apps' Resizer)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols, cn = img.shape
    dr, dc = resize_dims(rows, cols, scale)
    out = np.zeros((dr, dc, cn), dtype=np.uint8)
    if interp is None:
        interp = 3 if np.float32(scale) < np.float32(1.0) else 1
    rc = lib().acfo_resize_u8(img.ctypes.data, rows, cols, cn, 0, C.c_double(scale), C.c_double(scale), interp, out.ctypes.data, dr, dc)
    if rc:
        raise RuntimeError("acfo_resize_u8 rc=%d" % rc)
    return out


def unscale_rect(scale, box):
    a = (C.c_int * 4)(*[int(v) for v in box])
    o = (C.c_int * 4)()
    lib().acfo_unscale_rect(C.c_float(scale), a, o)
    return list(o)


def thrs_u8(thrs):
    t = np.ascontiguousarray(thrs, dtype=np.float32)
    out = np.zeros(t.shape, np.uint8)
    lib().acfo_thrs_u8(F(t), t.size, out.ctypes.data)
    return out


def acf_detect1_u8(plan, chns_u8, thrs_u8_arr, cap=1 << 16):
    """The uint8_t cascade body on one level's byte planes [nC][wP][hP] (acfo_acf_detect1 with u8 = 1)."""
    chns_u8 = np.ascontiguousarray(chns_u8, dtype=np.uint8)
    t = np.ascontiguousarray(thrs_u8_arr, dtype=np.uint8)
    nC, wP, hP = chns_u8.shape
    hits = np.zeros(cap, dtype=capi.HIT_DTYPE)
    n = lib().acfo_acf_detect1(chns_u8.ctypes.data, 1, t.ctypes.data, hP, wP, nC, C.byref(plan.params),
                               hits.ctypes.data_as(C.POINTER(capi.Hit)), cap, 0)
    return hits[:min(n, cap)].copy(), n


def mean_trees(plan, pyr):
    o = lib()
    lv = plan.levels
    tot, cnt = 0.0, 0
    for i in range(plan.nScales):
        nw = max(lv[i].nWinR, 0) * max(lv[i].nWinC, 0)
        if nw == 0:
            continue
        v = plan.level_view(pyr, i)
        tot += o.acfo_mean_trees(F(v), lv[i].hP, lv[i].wP, plan.nChns, C.byref(plan.params)) * nw
        cnt += nw
    return tot / max(cnt, 1), cnt
