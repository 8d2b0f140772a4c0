"""Thin Python handle on the C ABI (include/acf_hip.h) for tests and bench.py.

Every method is a direct call into libacf_hip.so; there is no Python or CPU
implementation behind it.  Frames are float32 [n][d][W][H] (H contiguous, the
reference's transposed planar layout, MatP.cpp:51-73) either as a torch CUDA
tensor / raw device pointer (run, pyramid) or as a numpy array (run_host).
"""
import ctypes as C

import numpy as np

from . import capi


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("acf_hip error %d: %s" % (code, msg))
        self.code = code


def _dev_ptr(x):
    if isinstance(x, int):
        return x
    # torch tensor
    assert x.is_cuda and x.is_contiguous() and x.dtype.is_floating_point and x.element_size() == 4
    return x.data_ptr()


def _dev_ptr_u8(x):
    if isinstance(x, int):
        return x
    assert x.is_cuda and x.is_contiguous() and x.element_size() == 1
    return x.data_ptr()


class PinnedBuffer:
    """Page-locked host memory from the library (acf_hip_host_alloc), viewed as a numpy uint8 array."""

    def __init__(self, nbytes):
        self.lib = capi.load()
        self.ptr = C.c_void_p()
        if self.lib.acf_hip_host_alloc(nbytes, C.byref(self.ptr)):
            raise HipError(5, "acf_hip_host_alloc failed")
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def close(self):
        if self.ptr:
            self.array = None
            self.lib.acf_hip_host_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipDetector:
    def __init__(self, model=None, H=0, W=0, d_in=3, max_batch=1, max_hits=4096, device=0, stream=None, taps=False, streams=1):
        self.lib = capi.load()
        self.ctx = C.c_void_p()
        rc = self.lib.acf_hip_create(device, C.c_void_p(stream or 0), C.byref(self.ctx))
        if rc:
            raise HipError(rc, "acf_hip_create failed (no gfx950 device?)")
        self._keep = None
        self.levels = []
        self.nChns = 0
        if taps:
            self._chk(self.lib.acf_hip_set_option(self.ctx, b"taps", 1))
        if streams > 1:
            # sub-batch contexts on their own streams (takes effect at plan time)
            self._chk(self.lib.acf_hip_set_option(self.ctx, b"streams", streams))
        if model is not None:
            self.set_model(model)
            if H and W:
                self.plan(H, W, d_in, max_batch, max_hits)

    def _chk(self, rc):
        if rc:
            raise HipError(rc, (self.lib.acf_hip_last_error(self.ctx) or b"").decode())

    def close(self):
        if self.ctx:
            self.lib.acf_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, model):
        self.model = model
        self.params, self._keep = capi.make_params(model)
        self._chk(self.lib.acf_hip_set_model(self.ctx, C.byref(self.params)))

    def plan(self, H, W, d_in=3, max_batch=1, max_hits=4096):
        self._chk(self.lib.acf_hip_plan(self.ctx, H, W, d_in, max_batch, max_hits))
        self.H, self.W, self.d_in, self.max_batch, self.max_hits = H, W, d_in, max_batch, max_hits
        n, nc = C.c_int(), C.c_int()
        self._chk(self.lib.acf_hip_num_levels(self.ctx, C.byref(n), C.byref(nc)))
        lv = (capi.Level * n.value)()
        self._chk(self.lib.acf_hip_get_levels(self.ctx, lv, n.value))
        self.levels = list(lv)
        self.nChns = nc.value
        tot = C.c_int64()
        self._chk(self.lib.acf_hip_pyramid_floats(self.ctx, C.byref(tot)))
        self.pyr_floats = tot.value
        self.ldcf_levels = []
        if int(self.params.ldcfK) > 0:
            ll = (capi.Level * n.value)()
            self._chk(self.lib.acf_hip_get_ldcf_levels(self.ctx, ll, n.value))
            self.ldcf_levels = list(ll)

    # ---- hot path
    def pyramid(self, frames, n=None):
        n = n if n is not None else frames.shape[0]
        self._chk(self.lib.acf_hip_pyramid(self.ctx, C.c_void_p(_dev_ptr(frames)), n))

    def detect(self):
        self._chk(self.lib.acf_hip_detect(self.ctx))

    def run(self, frames, n=None):
        n = n if n is not None else frames.shape[0]
        self._chk(self.lib.acf_hip_run(self.ctx, C.c_void_p(_dev_ptr(frames)), n))

    def run_host(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        self._chk(self.lib.acf_hip_run_host(self.ctx, capi.fptr(frames), frames.shape[0]))

    def pyramid_u8(self, frames, pix=capi.PIX_RGB, row_stride=0, n=None):
        """frames: torch uint8 CUDA tensor [n][H][W][cpp] (upright, packed) or a raw device pointer."""
        n = n if n is not None else frames.shape[0]
        self._chk(self.lib.acf_hip_pyramid_u8(self.ctx, C.c_void_p(_dev_ptr_u8(frames)), n, pix, row_stride))

    def run_u8(self, frames, pix=capi.PIX_RGB, row_stride=0, n=None):
        n = n if n is not None else frames.shape[0]
        self._chk(self.lib.acf_hip_run_u8(self.ctx, C.c_void_p(_dev_ptr_u8(frames)), n, pix, row_stride))

    # ---- the apps' resize to a minimum object width (acf.cpp:117-148 Resizer) in front of the 8-bit entries
    @staticmethod
    def resize_scale(win_width, min_width):
        """scale = float(winSize.width) / float(minWidth), a float (acf.cpp:124)."""
        return float(np.float32(win_width) / np.float32(min_width))

    @staticmethod
    def resize_dims(rows, cols, scale):
        r, c_ = C.c_int(), C.c_int()
        rc = capi.load().acf_hip_resize_dims(rows, cols, float(scale), C.byref(r), C.byref(c_))
        if rc:
            raise HipError(rc, "acf_hip_resize_dims")
        return r.value, c_.value

    def set_input_resize(self, src_rows, src_cols, scale):
        """The plan must be for resize_dims(src_rows, src_cols, scale); the 8-bit entries then take src_rows x src_cols frames."""
        self._chk(self.lib.acf_hip_set_input_resize(self.ctx, src_rows, src_cols, float(scale)))

    def op_resize_u8(self, img, scale):
        """img: uint8 [rows][cols][cpp] (host) -> the reduced image (cv::resize as the apps' Resizer calls it)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols, cpp = img.shape
        dr, dc = self.resize_dims(rows, cols, scale)
        out = np.zeros((dr, dc, cpp), dtype=np.uint8)
        self._chk(self.lib.acf_hip_op_resize_u8(self.ctx, img.ctypes.data, rows, cols, cpp, 0, float(scale), out.ctypes.data, dr, dc))
        return out

    @staticmethod
    def unscale_boxes(dets, scale):
        """Resizer::operator()(objects): cv::Rect2f(o) * (1.f / scale) -> cv::Rect (float products, cvRound); dets: DET_DTYPE array."""
        out = dets.copy()
        inv = np.float32(1.0) / np.float32(scale)
        for k in ("x", "y", "w", "h"):
            out[k] = np.rint((dets[k].astype(np.float32) * inv).astype(np.float64)).astype(out[k].dtype)
        return out

    # ---- streaming front end (pinned host frames in, pinned host records out)
    def stream_open(self, pix=capi.PIX_RGB, row_stride=0, cap=1024, depth=2):
        self._chk(self.lib.acf_hip_stream_open(self.ctx, pix, row_stride, cap, depth))
        self._stream_cap = cap

    def stream_submit(self, host_ptr, n):
        t = C.c_int()
        self._chk(self.lib.acf_hip_stream_submit(self.ctx, C.c_void_p(host_ptr), n, C.byref(t)))
        return t.value

    def stream_collect(self, ticket):
        """-> int32 array [n][1 + 6*cap] (a copy of the pinned records: count, then {x,y,w,h,score bits,scale})."""
        rec = C.POINTER(C.c_int32)()
        n = C.c_int()
        self._chk(self.lib.acf_hip_stream_collect(self.ctx, ticket, C.byref(rec), C.byref(n)))
        per = 1 + 6 * self._stream_cap
        return np.ctypeslib.as_array(rec, shape=(n.value, per)).copy()

    def stream_close(self):
        self._chk(self.lib.acf_hip_stream_close(self.ctx))

    def synchronize(self):
        self._chk(self.lib.acf_hip_synchronize(self.ctx))

    def repairs(self):
        """(smoothing planes checked, recomputed, level planes checked, recomputed) with option count_repairs on."""
        out = (C.c_int64 * 4)()
        self._chk(self.lib.acf_hip_get_repairs(self.ctx, out))
        return tuple(int(x) for x in out)

    def set_option(self, key, value):
        self._chk(self.lib.acf_hip_set_option(self.ctx, key.encode(), int(value)))

    def set_x86_tables(self, rcp, rsqrt):
        """Install one x86 CPU's _mm_rcp_ps / _mm_rsqrt_ps tables (4096 + 2 x 4096 uint32, acf_hip_set_x86_tables); option "arith" = 1
        then makes the reference's three approximate sites return that CPU's bits."""
        rcp = np.ascontiguousarray(rcp, dtype=np.uint32)
        rsqrt = np.ascontiguousarray(rsqrt, dtype=np.uint32)
        assert rcp.shape == (4096,) and rsqrt.shape == (8192,)
        u32p = C.POINTER(C.c_uint32)
        self._chk(self.lib.acf_hip_set_x86_tables(self.ctx, rcp.ctypes.data_as(u32p), rsqrt.ctypes.data_as(u32p)))

    def selftest_x86(self, first, count, stride=1):
        """(rcp digest, rsqrt digest, mismatches of gradMag's one-read form) of the device's table functions over first + i * stride."""
        out = (C.c_uint64 * 3)()
        self._chk(self.lib.acf_hip_selftest_x86(self.ctx, C.c_uint32(first), C.c_uint64(count), C.c_uint32(stride), out))
        return int(out[0]), int(out[1]), int(out[2])

    def profile(self):
        """{kernel name: (total ms, launches)} since the last call (option "profile")."""
        cap = 64
        n = C.c_int()
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        cnt = (C.c_int * cap)()
        self._chk(self.lib.acf_hip_profile_get(self.ctx, C.byref(n), names, ms, cnt, cap))
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(min(n.value, cap))}

    def export_detections(self, dst, cap):
        self._chk(self.lib.acf_hip_export_detections(self.ctx, C.c_void_p(dst.data_ptr()), cap))

    def detections(self, frame):
        det = np.zeros(self.max_hits, dtype=capi.DET_DTYPE)
        hits = np.zeros(self.max_hits, dtype=capi.HIT_DTYPE)
        n = C.c_int()
        self._chk(self.lib.acf_hip_get_detections(self.ctx, frame, det.ctypes.data_as(C.POINTER(capi.Detection)), self.max_hits, C.byref(n)))
        self._chk(self.lib.acf_hip_get_hits(self.ctx, frame, hits.ctypes.data_as(C.POINTER(capi.Hit)), self.max_hits, C.byref(n)))
        return det[:n.value].copy(), hits[:n.value].copy()

    def raw_detections(self, frame):
        """acfDetect1's list (scale, column, row order) whether or not the device NMS is on."""
        det = np.zeros(self.max_hits, dtype=capi.DET_DTYPE)
        n = C.c_int()
        self._chk(self.lib.acf_hip_get_raw_detections(self.ctx, frame, det.ctypes.data_as(C.POINTER(capi.Detection)), self.max_hits, C.byref(n)))
        return det[:n.value].copy()

    # ---- parity taps
    def set_nms(self, params):
        """params: capi.NmsParams (capi.make_nms) or None.  From the next detect()/run() on, detections() and
        export_detections() return the survivors of bbNms (+ prune), in score order."""
        self._nms_keep = params
        self._chk(self.lib.acf_hip_set_nms(self.ctx, C.byref(params) if params is not None else None))

    def op_nms(self, boxes, scores, params):
        """bbNms + prune of one host list: boxes int32 [n][4] = x, y, w, h; scores float64 [n] -> indices of the survivors in order."""
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        n = len(scores)
        keep = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int(0)
        self._chk(self.lib.acf_hip_op_nms(self.ctx, boxes.ctypes.data_as(C.POINTER(C.c_int32)), scores.ctypes.data_as(C.POINTER(C.c_double)), n,
                                          C.byref(params), keep.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(cnt)))
        return keep[:cnt.value].copy()

    def lambdas(self, frame=0):
        """The three lambdas frame `frame` of the last pyramid was approximated with (the model's, or estimated from the image)."""
        out = (C.c_double * 3)()
        self._chk(self.lib.acf_hip_get_lambdas(self.ctx, frame, out))
        return [out[0], out[1], out[2]]

    def read_level(self, frame, level):
        l = self.levels[level]
        out = np.zeros((self.nChns, l.wP, l.hP), dtype=np.float32)
        self._chk(self.lib.acf_hip_read_level(self.ctx, frame, level, capi.fptr(out)))
        return out

    def read_rank_level(self, frame, level):
        """The 16-bit threshold-rank cells of one level, as the cascade read them (after detect()/run())."""
        l = self.levels[level]
        out = np.zeros((self.nChns, l.wP, l.hP), dtype=np.uint16)
        self._chk(self.lib.acf_hip_read_rank_level(self.ctx, frame, level, out.ctypes.data_as(C.POINTER(C.c_uint16))))
        return out

    def read_pyramid(self, frame):
        return np.concatenate([self.read_level(frame, i).ravel() for i in range(len(self.levels))])

    def read_tap(self, frame, tap, index, shape):
        out = np.zeros(shape, dtype=np.float32)
        self._chk(self.lib.acf_hip_read_tap(self.ctx, frame, tap, index, capi.fptr(out), out.size))
        return out

    # ---- single operators (host planes in, host planes out)
    def op_rgb_convert(self, a, flag):
        a = np.ascontiguousarray(a, dtype=np.float32)
        _, w, h = a.shape
        out = np.zeros((1 if flag == capi.CS_GRAY else 3, w, h), dtype=np.float32)
        self._chk(self.lib.acf_hip_op_rgb_convert(self.ctx, capi.fptr(a), capi.fptr(out), h, w, flag))
        return out

    def op_conv_tri(self, a, r, aliased=True):
        a = np.ascontiguousarray(a, dtype=np.float32)
        d, w, h = a.shape
        out = np.zeros_like(a)
        self._chk(self.lib.acf_hip_op_conv_tri(self.ctx, capi.fptr(a), capi.fptr(out), h, w, d, float(r), int(aliased)))
        return out

    def op_gradient_mag(self, a, normRad=0, normConst=0.005, full=0):
        a = np.ascontiguousarray(a, dtype=np.float32)
        w, h = a.shape
        M, O, S = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
        self._chk(self.lib.acf_hip_op_gradient_mag(self.ctx, capi.fptr(a), capi.fptr(M), capi.fptr(O), capi.fptr(S), h, w, normRad, normConst, full))
        return M, O, S

    def chns_compute(self, a, model=None):
        """Detector::chnsCompute: a = host planes [d][w][h]; model = a dict like synth.make_model's (its Chns fields are read) or
        None for the context's model.  Returns [nChns][w / shrink][h / shrink]."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        d, w, h = a.shape
        pp, keep = (None, None)
        if model is not None:
            prm, keep = capi.make_params(model)
            pp = C.byref(prm)
        n, hc, wc = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.acf_hip_chns_compute(self.ctx, pp, capi.fptr(a), h, w, d, None, 0, C.byref(n), C.byref(hc), C.byref(wc)))
        out = np.zeros((n.value, wc.value, hc.value), np.float32)
        self._chk(self.lib.acf_hip_chns_compute(self.ctx, pp, capi.fptr(a), h, w, d, capi.fptr(out), out.size, C.byref(n), C.byref(hc), C.byref(wc)))
        return out

    def op_gradient_hist(self, M, O, bin=4, nOrients=6, full=0, softBin=0):
        M = np.ascontiguousarray(M, dtype=np.float32)
        O = np.ascontiguousarray(O, dtype=np.float32)
        w, h = M.shape
        H = np.zeros((nOrients, w // bin, h // bin), dtype=np.float32)
        self._chk(self.lib.acf_hip_op_gradient_hist(self.ctx, capi.fptr(M), capi.fptr(O), capi.fptr(H), h, w, bin, nOrients, softBin, full))
        return H

    def op_im_resample(self, a, hb, wb, nrm=1.0):
        a = np.ascontiguousarray(a, dtype=np.float32)
        d, wa, ha = a.shape
        out = np.zeros((d, wb, hb), dtype=np.float32)
        self._chk(self.lib.acf_hip_op_im_resample(self.ctx, capi.fptr(a), capi.fptr(out), ha, wa, hb, wb, d, float(nrm)))
        return out

    def op_acf_detect1(self, chns, cap=1 << 16):
        chns = np.ascontiguousarray(chns, dtype=np.float32)
        nC, wP, hP = chns.shape
        hits = np.zeros(cap, dtype=capi.HIT_DTYPE)
        n = C.c_int()
        self._chk(self.lib.acf_hip_op_acf_detect1(self.ctx, capi.fptr(chns), hP, wP, nC, hits.ctypes.data_as(C.POINTER(capi.Hit)), cap, C.byref(n)))
        return hits[:n.value].copy()

    def op_acf_detect1_u8(self, chns, thrs_u8=None, cap=1 << 16):
        """uint8 channel planes [nC][wP][hP]; thrs_u8 None: derived from the model's thrs (ACFIOArchive.h:96-99)."""
        chns = np.ascontiguousarray(chns, dtype=np.uint8)
        nC, wP, hP = chns.shape
        hits = np.zeros(cap, dtype=capi.HIT_DTYPE)
        n = C.c_int()
        t = None if thrs_u8 is None else np.ascontiguousarray(thrs_u8, dtype=np.uint8)
        self._chk(self.lib.acf_hip_op_acf_detect1_u8(self.ctx, chns.ctypes.data, hP, wP, nC, None if t is None else t.ctypes.data,
                                                     hits.ctypes.data_as(C.POINTER(capi.Hit)), cap, C.byref(n)))
        return hits[:n.value].copy()

    def op_evaluate(self, chns, cascThr=0.0):
        """Detector::evaluate: score of the window at (0, 0) of [nC][wP][hP] channels (acfDetect1.cpp:337-342)."""
        chns = np.ascontiguousarray(chns, dtype=np.float32)
        nC, wP, hP = chns.shape
        out = np.zeros(1, np.float32)
        self._chk(self.lib.acf_hip_op_evaluate(self.ctx, capi.fptr(chns), hP, wP, nC, float(cascThr), capi.fptr(out)))
        return out[0]


class DetectorPool:
    """N detector contexts on one GPU, each with its own HIP stream, plan and buffers; batches are handed to them in turn.

    The reference runs one acf::Detector per thread (src/app/acf/acf.cpp:255-320); here the unit is a stream.  The hot
    path alternates HBM-bound kernels (smoothing, gradMag, running sums) and VALU/LDS-bound ones (level chains, cascade), so
    independent streams let one batch's cascade run while another batch's pyramid waits on memory: 3 contexts x 96 frames
    measure +16 % over one context x 256 frames on an MI355X (profiles/ubench/two_contexts.py).
    """

    def __init__(self, n, model, H, W, d_in=3, max_batch=1, max_hits=4096, device=0, shared_device=True, **kw):
        """shared_device (default on, n > 1 only): the contexts' kernel forms are chosen for the least total work because other
        contexts' kernels fill the machine meanwhile (uncut smoothing chains, convTri's x pass on the gradient plane's chain for
        batches >= 64 frames): +4.3 % frames/s when the contexts really run side by side, -7 % when they do not — a pool that is
        fed one context at a time with a synchronise between batches should pass shared_device=False."""
        import torch
        self.streams = [torch.cuda.Stream(device=torch.device("cuda", device)) for _ in range(n)]
        self.dets = [HipDetector(model, H, W, d_in, max_batch=max_batch, max_hits=max_hits, device=device, stream=s.cuda_stream, **kw)
                     for s in self.streams]
        if n > 1:
            # the contexts' level and tile kernels (VALU / LDS-bound) take turns, so that each runs beside the other contexts'
            # memory-bound pyramid kernels and not beside another of its kind (acf_hip.h, option cascade_turns): +4 % frames/s
            # and the tile kernel runs one workgroup per tile: persistent workgroups hold every CU's LDS for the whole kernel and
            # keep the other contexts' kernels out (3 contexts: 14.0k against 13.4k frames/s; alone it is the other way round)
            # and kernel forms are chosen for the least work, not the shortest time alone (option shared_device: the smoothing chains
            # uncut, convTri's x pass on the gradient plane's chain; 3 contexts x 96 frames: +4.3 % frames/s)
            for d in self.dets:
                d.set_option("cascade_turns", 5)
                d.set_option("tile_persist", 0)
                d.set_option("shared_device", 1 if shared_device else 0)

    def __len__(self):
        return len(self.dets)

    def __iter__(self):
        return iter(zip(self.dets, self.streams))

    def synchronize(self):
        for d in self.dets:
            d.synchronize()

    def close(self):
        for d in self.dets:
            d.close()
