"""Writer of the "ACFHIPM1" model file read by the C++ host CLI
(acf_amd/host/acf_hip_detect.cpp): a text header of "key value" lines, "END",
then raw little-endian fids u32, thrs f32, hs f32, child u32 arrays
([nTrees][nTreeNodes] row-major, the layout after the reference's load-time
transpose, ACFIO.cpp:61-67)."""
import numpy as np

_KEYS = ("treeDepth", "modelDs_h", "modelDs_w", "modelDsPad_h", "modelDsPad_w", "stride", "cascThr", "nPerOct", "nOctUp",
         "nApprox", "pad_h", "pad_w", "minDs_h", "minDs_w", "smooth", "shrink", "colorEnabled", "colorSmooth",
         "colorSpace", "gradMagEnabled", "colorChn", "normRad", "normConst", "full", "gradHistEnabled", "binSize",
         "nOrients", "softBin")


def write_model(path, model):
    fids = np.ascontiguousarray(model["fids"], dtype="<u4")
    nT, nN = fids.shape
    with open(path, "wb") as f:
        f.write(b"ACFHIPM1\n")
        f.write(("nTrees %d\nnTreeNodes %d\n" % (nT, nN)).encode())
        for k in _KEYS:
            f.write(("%s %r\n" % (k, model[k])).encode())
        f.write(("lambdas %s\n" % " ".join(repr(float(v)) for v in (model.get("lambdas") or []))).encode())
        f.write(b"END\n")
        f.write(fids.tobytes())
        f.write(np.ascontiguousarray(model["thrs"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(model["hs"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(model["child"], dtype="<u4").tobytes())
