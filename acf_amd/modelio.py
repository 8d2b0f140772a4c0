"""Writer of the "ACFHIPM1" model file read by the C++ host CLI
(acf_amd/host/acf_hip_detect.cpp): a text header of "key value" lines, "END",
then raw little-endian fids u32, thrs f32, hs f32, child u32 arrays
([nTrees][nTreeNodes] row-major, the layout after the reference's load-time
transpose, ACFIO.cpp:61-67); with "ldcfK"/"ldcfCount" header keys, the LDCF
filters [k][nChns][5][5] f32 follow."""
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
        filt = model.get("ldcfFilters") if int(model.get("ldcfK", 0)) > 0 else None
        if filt is not None:
            f.write(("ldcfK %d\nldcfCount %d\n" % (int(model["ldcfK"]), int(np.asarray(filt).size))).encode())
        f.write(b"END\n")
        f.write(fids.tobytes())
        f.write(np.ascontiguousarray(model["thrs"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(model["hs"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(model["child"], dtype="<u4").tobytes())
        if filt is not None:
            f.write(np.ascontiguousarray(filt, dtype="<f4").tobytes())  # LDCF filters [k][nChns][5][5] after the tree arrays


# ---------------------------------------------------------------------------
# `.cpb`: acf::Detector through cereal::PortableBinaryOutputArchive
# (src/lib/acf/io/cereal_pba.h:66-79; field order ACFIOArchive.h:75-216; Field<T> ACFField.h:123-130;
# cv::Mat io/cvmat_cereal.h:20-73).  A second, independent restatement of the layout documented in
# acf_amd/host/ModelIO.h — the C++ reader/writer is tested against this one byte for byte.
# ---------------------------------------------------------------------------
import struct

_CS = ("gray", "rgb", "luv", "hsv", "orig")


class _Out:
    def __init__(self):
        self.b = bytearray([1])  # little-endian flag
        self.seen = set()

    def ver(self, t, v=0):
        if t not in self.seen:
            self.seen.add(t)
            self.b += struct.pack("<I", v)

    def i32(self, v):
        self.b += struct.pack("<i", int(v))

    def f64(self, v):
        self.b += struct.pack("<d", float(v))

    def flag(self, v):
        self.b += bytes([1 if v else 0])

    def string(self, s):
        s = s.encode()
        self.b += struct.pack("<Q", len(s)) + s

    def meta(self, name, leaf):
        self.string(name)
        self.flag(True)
        self.flag(leaf)

    def f_int(self, name, v):
        self.ver("Field<int>")
        self.i32(v)
        self.meta(name, True)

    def f_double(self, name, v):
        self.ver("Field<double>")
        self.f64(v)
        self.meta(name, True)

    def f_string(self, name, v):
        self.ver("Field<string>")
        self.string(v)
        self.meta(name, True)

    def f_size(self, name, w, h):
        self.ver("Field<Size>")
        self.ver("cv::Size")
        self.i32(w)
        self.i32(h)
        self.meta(name, True)

    def f_vec(self, name, t, fmt, v):
        self.ver(t)
        self.b += struct.pack("<Q", len(v))
        for e in v:
            self.b += struct.pack(fmt, e)
        self.meta(name, True)

    def mat(self, a, cvtype):
        self.ver("cv::Mat")
        self.i32(a.shape[0])
        self.i32(a.shape[1])
        self.i32(cvtype)
        self.flag(True)
        self.b += a.tobytes()


def write_cpb(path, model, nms=("maxg", 0.65, "min"), cascCal=0.0):
    o = _Out()
    fids = np.ascontiguousarray(model["fids"]).astype("<i4")
    nT, nN = fids.shape
    o.ver("Detector", 1)
    o.ver("Classifier")
    o.mat(fids, 4)
    o.mat(np.ascontiguousarray(model["thrs"], dtype="<f4").reshape(nT, nN), 5)
    o.mat(np.ascontiguousarray(model["child"]).astype("<i4").reshape(nT, nN), 4)
    o.mat(np.ascontiguousarray(model["hs"], dtype="<f4").reshape(nT, nN), 5)
    o.mat(np.zeros((nT, nN), "<f4"), 5)  # weights
    level = np.floor(np.log2(np.arange(nN) + 1)).astype("<i4")
    o.mat(np.ascontiguousarray(np.broadcast_to(level, (nT, nN))), 4)  # depth
    o.b += struct.pack("<Q", 0) + struct.pack("<Q", 0)  # errs, losses
    o.i32(model["treeDepth"])
    o.ver("Options")
    o.ver("Field<Pyramid>"); o.ver("Pyramid")
    o.ver("Field<Chns>"); o.ver("Chns")
    o.f_int("shrink", model["shrink"])
    o.f_int("complete", 1)
    o.ver("Field<Color>"); o.ver("Color")
    o.f_int("enabled", model["colorEnabled"])
    o.f_double("smooth", model["colorSmooth"])
    o.f_string("colorSpace", _CS[model["colorSpace"]])
    o.meta("pColor", False)
    o.ver("Field<GradMag>"); o.ver("GradMag")
    o.f_int("enabled", model["gradMagEnabled"])
    o.f_int("colorChn", model["colorChn"])
    o.f_int("normRad", model["normRad"])
    o.f_double("normConst", model["normConst"])
    o.f_int("full", model["full"])
    o.meta("pGradMag", False)
    o.ver("Field<GradHist>"); o.ver("GradHist")
    o.f_int("enabled", model["gradHistEnabled"])
    o.f_int("binSize", model["binSize"])
    o.f_int("nOrients", model["nOrients"])
    o.f_int("softBin", model["softBin"])
    o.f_int("useHog", 0)
    o.f_double("clipHog", 0.2)
    o.meta("pGradHist", False)
    o.meta("pChns", False)
    o.f_int("nPerOct", model["nPerOct"])
    o.f_int("nOctUp", model["nOctUp"])
    o.f_int("nApprox", model["nApprox"])
    o.f_vec("lambdas", "Field<vector<double>>", "<d", [float(v) for v in (model.get("lambdas") or [])])
    o.f_size("pad", model["pad_h"], model["pad_w"])        # cv::Size{width = image-height axis} (ACFIO.h:168-181)
    o.f_size("minDs", model["minDs_h"], model["minDs_w"])
    o.f_double("smooth", model["smooth"])
    o.f_int("concat", 1)
    o.f_int("complete", 1)
    o.meta("pPyramid", False)
    o.f_size("modelDs", model["modelDs_h"], model["modelDs_w"])
    o.f_size("modelDsPad", model["modelDsPad_h"], model["modelDsPad_w"])
    o.ver("Field<Nms>"); o.ver("Nms")
    o.f_string("type", nms[0])
    o.f_double("overlap", nms[1])
    o.f_string("ovrDnm", nms[2])
    o.meta("pNms", False)
    o.f_int("stride", model["stride"])
    o.f_double("cascThr", model["cascThr"])
    o.f_double("cascCal", cascCal)
    o.f_vec("nWeak", "Field<vector<int>>", "<i", [])
    o.ver("Field<Boost>"); o.ver("Boost"); o.ver("Field<Tree>"); o.ver("Tree")
    o.f_int("nBins", 256)
    o.f_int("maxDepth", 2)
    o.f_double("minWeight", 0.01)
    o.f_double("fracFtrs", 1)
    o.f_int("nThreads", 16)
    o.meta("pTree", False)
    o.f_int("nWeak", 128)
    o.f_int("discrete", 1)
    o.f_int("verbose", 16)
    o.meta("pBoost", False)
    for k in ("posGtDir", "posImgDir", "negImgDir", "posWinDir", "negWinDir"):
        o.f_string(k, "")
    for k in ("nPos", "nNeg", "nPerNeg", "nAccNeg"):
        o.f_int(k, 0)
    o.ver("Field<Jitter>"); o.ver("Jitter")
    o.f_int("flip", 0)
    o.meta("pJitter", False)
    o.f_int("winsSave", 0)
    with open(path, "wb") as f:
        f.write(bytes(o.b))


class _In:
    def __init__(self, data):
        self.d, self.p, self.seen = data, 1, set()
        if data[0] != 1:
            raise ValueError("cpb: only little-endian streams are read here")

    def take(self, fmt):
        n = struct.calcsize(fmt)
        v = struct.unpack_from(fmt, self.d, self.p)
        self.p += n
        return v[0]

    def ver(self, t):
        if t not in self.seen:
            self.seen.add(t)
            return self.take("<I")
        return None

    def string(self):
        n = self.take("<Q")
        s = self.d[self.p:self.p + n].decode()
        self.p += n
        return s

    def meta(self):
        return self.string(), bool(self.take("<B")), bool(self.take("<B"))

    def field(self, t, reader):
        self.ver(t)
        v = reader()
        name, has, leaf = self.meta()
        return v if has else None

    def f_int(self):
        return self.field("Field<int>", lambda: self.take("<i"))

    def f_double(self):
        return self.field("Field<double>", lambda: self.take("<d"))

    def f_string(self):
        return self.field("Field<string>", self.string)

    def f_size(self):
        def rd():
            self.ver("cv::Size")
            return (self.take("<i"), self.take("<i"))
        return self.field("Field<Size>", rd)

    def f_vec(self, t, fmt):
        return self.field(t, lambda: [self.take(fmt) for _ in range(self.take("<Q"))])

    def mat(self):
        self.ver("cv::Mat")
        r, c, t, cont = self.take("<i"), self.take("<i"), self.take("<i"), self.take("<B")
        dt = {4: "<i4", 5: "<f4", 6: "<f8", 0: "u1"}[t & 7]
        a = np.frombuffer(self.d, dtype=dt, count=r * c, offset=self.p).reshape(r, c).copy()
        self.p += a.nbytes
        return a


def read_cpb(path):
    """-> (model dict with the keys write_model uses, nms tuple, cascCal)."""
    i = _In(open(path, "rb").read())
    m = {}
    assert i.ver("Detector") == 1
    i.ver("Classifier")
    m["fids"] = i.mat().astype(np.uint32)
    m["thrs"] = i.mat()
    m["child"] = i.mat().astype(np.uint32)
    m["hs"] = i.mat()
    i.mat()
    i.mat()
    for _ in range(2):
        n = i.take("<Q")
        i.p += 8 * n
    m["treeDepth"] = i.take("<i")
    i.ver("Options")
    i.ver("Field<Pyramid>"); i.ver("Pyramid"); i.ver("Field<Chns>"); i.ver("Chns")
    m["shrink"] = i.f_int()
    i.f_int()
    i.ver("Field<Color>"); i.ver("Color")
    m["colorEnabled"], m["colorSmooth"] = i.f_int(), i.f_double()
    m["colorSpace"] = _CS.index(i.f_string())
    i.meta()
    i.ver("Field<GradMag>"); i.ver("GradMag")
    m["gradMagEnabled"], m["colorChn"], m["normRad"], m["normConst"], m["full"] = i.f_int(), i.f_int(), i.f_int(), i.f_double(), i.f_int()
    i.meta()
    i.ver("Field<GradHist>"); i.ver("GradHist")
    m["gradHistEnabled"], m["binSize"], m["nOrients"], m["softBin"] = i.f_int(), i.f_int(), i.f_int(), i.f_int()
    i.f_int(), i.f_double()
    i.meta()
    i.meta()
    m["nPerOct"], m["nOctUp"], m["nApprox"] = i.f_int(), i.f_int(), i.f_int()
    m["lambdas"] = i.f_vec("Field<vector<double>>", "<d")
    (m["pad_h"], m["pad_w"]), (m["minDs_h"], m["minDs_w"]) = i.f_size(), i.f_size()
    m["smooth"] = i.f_double()
    i.f_int(), i.f_int()
    i.meta()
    (m["modelDs_h"], m["modelDs_w"]), (m["modelDsPad_h"], m["modelDsPad_w"]) = i.f_size(), i.f_size()
    i.ver("Field<Nms>"); i.ver("Nms")
    nms = (i.f_string(), i.f_double(), i.f_string())
    i.meta()
    m["stride"], m["cascThr"] = i.f_int(), i.f_double()
    cal = i.f_double()
    i.f_vec("Field<vector<int>>", "<i")
    i.ver("Field<Boost>"); i.ver("Boost"); i.ver("Field<Tree>"); i.ver("Tree")
    i.f_int(), i.f_int(), i.f_double(), i.f_double(), i.f_int()
    i.meta()
    i.f_int(), i.f_int(), i.f_int()
    i.meta()
    for _ in range(5):
        i.f_string()
    for _ in range(4):
        i.f_int()
    i.ver("Field<Jitter>"); i.ver("Jitter")
    i.f_int()
    i.meta()
    i.f_int()
    if i.p != len(i.d):
        raise ValueError("cpb: %d trailing bytes" % (len(i.d) - i.p))
    return m, nms, cal
