"""ctypes binding of the C ABI declared in include/acf_hip.h.

This is plumbing for tests and bench.py; the product is libacf_hip.so itself.
There is no CPU fallback: if the HIP library has not been built, importing
`load()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACF_HIP_LIB: another build of the same library (profiles/build_variant.sh: A/B of compile-time constants on one box)
LIB_PATH = os.environ.get("ACF_HIP_LIB") or os.path.join(_HERE, "libacf_hip.so")

OK = 0
E_INVALID, E_UNSUPPORTED, E_NOMODEL, E_NOPLAN, E_HIP, E_NODEVICE, E_CAPACITY = 1, 2, 3, 4, 5, 6, 7  # include/acf_hip.h:44-51
NMS_CAP = 2048  # ACF_HIP_NMS_CAP
CS_GRAY, CS_RGB, CS_LUV, CS_HSV, CS_ORIG = 0, 1, 2, 3, 4
TAP_IMAGE, TAP_SMOOTHED, TAP_M, TAP_O, TAP_S, TAP_MNORM, TAP_CHNS, TAP_LDCF = range(8)


class Params(C.Structure):
    """struct acf_hip_params (include/acf_hip.h)."""

    _fields_ = [
        ("nTrees", C.c_int32),
        ("nTreeNodes", C.c_int32),
        ("treeDepth", C.c_int32),
        ("fids", C.POINTER(C.c_uint32)),
        ("thrs", C.POINTER(C.c_float)),
        ("hs", C.POINTER(C.c_float)),
        ("child", C.POINTER(C.c_uint32)),
        ("modelDs_h", C.c_int32),
        ("modelDs_w", C.c_int32),
        ("modelDsPad_h", C.c_int32),
        ("modelDsPad_w", C.c_int32),
        ("stride", C.c_int32),
        ("cascThr", C.c_double),
        ("nPerOct", C.c_int32),
        ("nOctUp", C.c_int32),
        ("nApprox", C.c_int32),
        ("nLambdas", C.c_int32),
        ("lambdas", C.c_double * 3),
        ("pad_h", C.c_int32),
        ("pad_w", C.c_int32),
        ("minDs_h", C.c_int32),
        ("minDs_w", C.c_int32),
        ("smooth", C.c_double),
        ("shrink", C.c_int32),
        ("colorEnabled", C.c_int32),
        ("colorSmooth", C.c_double),
        ("colorSpace", C.c_int32),
        ("gradMagEnabled", C.c_int32),
        ("colorChn", C.c_int32),
        ("normRad", C.c_int32),
        ("normConst", C.c_double),
        ("full", C.c_int32),
        ("gradHistEnabled", C.c_int32),
        ("binSize", C.c_int32),
        ("nOrients", C.c_int32),
        ("softBin", C.c_int32),
        ("isLuv", C.c_int32),
        ("ldcfK", C.c_int32),
        ("ldcfFilters", C.POINTER(C.c_float)),
    ]


class NmsParams(C.Structure):
    _fields_ = [("type", C.c_int32), ("ovrDnmUnion", C.c_int32), ("overlap", C.c_double), ("thr", C.c_double),
                ("prune", C.c_int32), ("maxCount", C.c_int32), ("pruneRatio", C.c_double)]


def make_nms(type="maxg", overlap=0.65, ovrDnm="min", thr=-1.7976931348623157e308, prune=False, maxCount=10, pruneRatio=0.0):
    """acf_hip_nms_params from the reference's option names (Options::Nms, ObjectDetector setters)."""
    return NmsParams({"none": 0, "max": 1, "maxg": 2}[type], 1 if ovrDnm == "union" else 0, float(overlap), float(thr),
                     1 if prune else 0, int(maxCount), float(pruneRatio))


class Detection(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("score", C.c_float), ("scale", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("scale", C.c_int32), ("c", C.c_int32), ("r", C.c_int32), ("score", C.c_float)]


class Level(C.Structure):
    _fields_ = [
        ("scale", C.c_double),
        ("scalehw_h", C.c_double),
        ("scalehw_w", C.c_double),
        ("isReal", C.c_int32),
        ("realIndex", C.c_int32),
        ("hC", C.c_int32),
        ("wC", C.c_int32),
        ("hP", C.c_int32),
        ("wP", C.c_int32),
        ("nWinR", C.c_int32),
        ("nWinC", C.c_int32),
        ("offset", C.c_int64),
    ]


DET_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("score", "<f4"), ("scale", "<i4")])
HIT_DTYPE = np.dtype([("scale", "<i4"), ("c", "<i4"), ("r", "<i4"), ("score", "<f4")])


def make_params(model):
    """Build a Params from a model dict (see acf_amd.synth.make_model).

    Returns (params, keepalive): the numpy arrays the pointers refer to must
    outlive every use of `params`.
    """
    p = Params()
    keep = {}
    for name, ctype, dtype in (("fids", C.c_uint32, np.uint32), ("thrs", C.c_float, np.float32),
                               ("hs", C.c_float, np.float32), ("child", C.c_uint32, np.uint32)):
        arr = model.get(name)
        if arr is None:
            setattr(p, name, C.POINTER(ctype)())
            continue
        arr = np.ascontiguousarray(arr, dtype=dtype)
        keep[name] = arr
        setattr(p, name, arr.ctypes.data_as(C.POINTER(ctype)))
    p.nTrees, p.nTreeNodes = keep["fids"].shape
    for k in ("treeDepth", "modelDs_h", "modelDs_w", "modelDsPad_h", "modelDsPad_w", "stride", "nPerOct", "nOctUp",
              "nApprox", "pad_h", "pad_w", "minDs_h", "minDs_w", "shrink", "colorEnabled", "colorSpace",
              "gradMagEnabled", "colorChn", "normRad", "full", "gradHistEnabled", "binSize", "nOrients", "softBin",
              "isLuv"):
        setattr(p, k, int(model[k]))
    for k in ("cascThr", "smooth", "colorSmooth", "normConst"):
        setattr(p, k, float(model[k]))
    filt = model.get("ldcfFilters")
    if filt is not None and int(model.get("ldcfK", 0)) > 0:
        filt = np.ascontiguousarray(filt, dtype=np.float32)  # [k][nChns][5 (dx)][5 (dy)]
        keep["ldcfFilters"] = filt
        p.ldcfK = int(model["ldcfK"])
        p.ldcfFilters = filt.ctypes.data_as(C.POINTER(C.c_float))
    lam = model.get("lambdas") or []
    p.nLambdas = len(lam)
    for i, v in enumerate(lam):
        p.lambdas[i] = float(v)
    return p, keep


PIX_RGB, PIX_BGR, PIX_RGBA, PIX_BGRA, PIX_GRAY = range(5)
PIX_CPP = {PIX_RGB: 3, PIX_BGR: 3, PIX_RGBA: 4, PIX_BGRA: 4, PIX_GRAY: 1}

_lib = None


def load():
    """dlopen libacf_hip.so and declare every entry point of include/acf_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    # A process that also uses torch must let torch load ITS bundled HIP runtime
    # first: if libacf_hip.so pulls in the system libamdhip64 before torch is
    # imported, torch later reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libacf_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    ctx = C.c_void_p
    fp = C.POINTER(C.c_float)
    sig = {
        "acf_hip_create": ([C.c_int, C.c_void_p, C.POINTER(ctx)], C.c_int),
        "acf_hip_destroy": ([ctx], C.c_int),
        "acf_hip_abi_version": ([], C.c_int),
        "acf_hip_device_count": ([C.POINTER(C.c_int)], C.c_int),
        "acf_hip_last_error": ([ctx], C.c_char_p),
        "acf_hip_set_option": ([ctx, C.c_char_p, C.c_int], C.c_int),
        "acf_hip_set_model": ([ctx, C.POINTER(Params)], C.c_int),
        "acf_hip_get_scales": ([C.c_int] * 7 + [C.POINTER(C.c_double)] * 3 + [C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_plan_levels": ([C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(Level), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_plan": ([ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_num_levels": ([ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_get_levels": ([ctx, C.POINTER(Level), C.c_int], C.c_int),
        "acf_hip_get_ldcf_levels": ([ctx, C.POINTER(Level), C.c_int], C.c_int),
        "acf_hip_pyramid_floats": ([ctx, C.POINTER(C.c_int64)], C.c_int),
        "acf_hip_get_lambdas": ([ctx, C.c_int, C.POINTER(C.c_double)], C.c_int),
        "acf_hip_pyramid": ([ctx, C.c_void_p, C.c_int], C.c_int),
        "acf_hip_detect": ([ctx], C.c_int),
        "acf_hip_run": ([ctx, C.c_void_p, C.c_int], C.c_int),
        "acf_hip_run_host": ([ctx, fp, C.c_int], C.c_int),
        "acf_hip_pyramid_u8": ([ctx, C.c_void_p, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_run_u8": ([ctx, C.c_void_p, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_resize_dims": ([C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_set_input_resize": ([ctx, C.c_int, C.c_int, C.c_double], C.c_int),
        "acf_hip_op_resize_u8": ([ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int], C.c_int),
        "acf_hip_stream_open": ([ctx, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_stream_submit": ([ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_stream_collect": ([ctx, C.c_int, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_stream_close": ([ctx], C.c_int),
        "acf_hip_host_alloc": ([C.c_size_t, C.POINTER(C.c_void_p)], C.c_int),
        "acf_hip_host_free": ([C.c_void_p], C.c_int),
        "acf_hip_set_nms": ([ctx, C.POINTER(NmsParams)], C.c_int),
        "acf_hip_op_nms": ([ctx, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.POINTER(NmsParams), C.POINTER(C.c_int32), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_get_detections": ([ctx, C.c_int, C.POINTER(Detection), C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_get_hits": ([ctx, C.c_int, C.POINTER(Hit), C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_get_raw_detections": ([ctx, C.c_int, C.POINTER(Detection), C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_export_detections": ([ctx, C.c_void_p, C.c_int], C.c_int),
        "acf_hip_synchronize": ([ctx], C.c_int),
        "acf_hip_get_repairs": ([ctx, C.POINTER(C.c_int64)], C.c_int),
        "acf_hip_profile_get": ([ctx, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int], C.c_int),
        "acf_hip_read_level": ([ctx, C.c_int, C.c_int, fp], C.c_int),
        "acf_hip_read_rank_level": ([ctx, C.c_int, C.c_int, C.POINTER(C.c_uint16)], C.c_int),
        "acf_hip_rank_cells_host": ([C.POINTER(Params), C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)], C.c_int),
        "acf_hip_read_tap": ([ctx, C.c_int, C.c_int, C.c_int, fp, C.c_int64], C.c_int),
        "acf_hip_op_rgb_convert": ([ctx, fp, fp, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_op_conv_tri": ([ctx, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int], C.c_int),
        "acf_hip_op_gradient_mag": ([ctx, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int], C.c_int),
        "acf_hip_selftest_gradmag": ([ctx, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)], C.c_int),
        "acf_hip_set_x86_tables": ([ctx, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)], C.c_int),
        "acf_hip_selftest_x86": ([ctx, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)], C.c_int),
        "acf_hip_chns_compute": ([ctx, C.POINTER(Params), fp, C.c_int, C.c_int, C.c_int, fp, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "acf_hip_op_gradient_hist": ([ctx, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
        "acf_hip_op_im_resample": ([ctx, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double], C.c_int),
        "acf_hip_op_acf_detect1": ([ctx, fp, C.c_int, C.c_int, C.c_int, C.POINTER(Hit), C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_op_acf_detect1_u8": ([ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Hit), C.c_int, C.POINTER(C.c_int)], C.c_int),
        "acf_hip_thrs_u8": ([fp, C.c_int, C.c_void_p], C.c_int),
        "acf_hip_op_evaluate": ([ctx, fp, C.c_int, C.c_int, C.c_int, C.c_double, fp], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    lib._declared = sorted(sig)
    _lib = lib
    return lib


DECLARED_SYMBOLS = [
    "acf_hip_create", "acf_hip_device_count", "acf_hip_destroy", "acf_hip_abi_version", "acf_hip_last_error", "acf_hip_set_option",
    "acf_hip_set_model", "acf_hip_get_scales", "acf_hip_plan_levels",
    "acf_hip_plan", "acf_hip_num_levels", "acf_hip_get_levels", "acf_hip_get_ldcf_levels", "acf_hip_pyramid_floats", "acf_hip_get_lambdas", "acf_hip_pyramid",
    "acf_hip_detect", "acf_hip_run", "acf_hip_run_host", "acf_hip_set_nms", "acf_hip_op_nms", "acf_hip_get_detections", "acf_hip_get_hits", "acf_hip_get_raw_detections",
    "acf_hip_pyramid_u8", "acf_hip_run_u8", "acf_hip_resize_dims", "acf_hip_set_input_resize", "acf_hip_op_resize_u8", "acf_hip_stream_open", "acf_hip_stream_submit", "acf_hip_stream_collect",
    "acf_hip_stream_close", "acf_hip_host_alloc", "acf_hip_host_free",
    "acf_hip_export_detections", "acf_hip_synchronize", "acf_hip_get_repairs", "acf_hip_profile_get", "acf_hip_read_level", "acf_hip_read_rank_level", "acf_hip_rank_cells_host", "acf_hip_read_tap",
    "acf_hip_op_rgb_convert", "acf_hip_op_conv_tri", "acf_hip_op_gradient_mag", "acf_hip_selftest_gradmag", "acf_hip_set_x86_tables", "acf_hip_selftest_x86", "acf_hip_chns_compute", "acf_hip_op_gradient_hist",
    "acf_hip_op_im_resample", "acf_hip_op_acf_detect1", "acf_hip_op_acf_detect1_u8", "acf_hip_thrs_u8", "acf_hip_op_evaluate",
]


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))
