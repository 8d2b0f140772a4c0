/*
 * acf_hip.h — C ABI of the MI355X (gfx950) ACF detection backend.
 *
 * This is the drop-in boundary for the chnsPyramid + acfDetect hot path of
 * elucideye/acf.  The reference has no C ABI of its own (its boundary is the
 * C++ class acf::Detector, src/lib/acf/acf/ACF.h:50-624); each entry point
 * below names the reference interface it stands in for.  The host-side C++
 * class in acf_amd/host/ (same method names as acf::Detector) dlopen()s this
 * library; tests and bench.py bind it with ctypes.
 *
 * Conventions
 *  - Every function returns an int status (ACF_HIP_OK == 0) and never throws.
 *    The reference returns 0 on the hot path and throws on precondition
 *    failures (ACF.cpp:135-141, wrappers.hpp:29-32); here those become codes.
 *  - A plane is float[w][h] with h (image height, image-y) contiguous: the
 *    reference's transposed planar layout (MatP, MatP.cpp:51-73; ACF.cpp:137).
 *    A frame is `d` planes back to back; a batch is frames back to back.
 *  - All sizes are given in upright-image terms (h = image height, w = image
 *    width).  The reference stores them in cv::Size with the members swapped
 *    ({width = h, height = w}, ACFIO.h:168-181); the host class does that
 *    translation, this ABI never sees cv::Size.
 *  - "dev" pointers are HIP device pointers; "host" pointers are ordinary
 *    memory.  No torch / OpenCV types appear in any signature.
 */
#ifndef ACF_HIP_H
#define ACF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACF_HIP_ABI_VERSION 9

#if defined(__GNUC__)
#define ACF_HIP_API __attribute__((visibility("default")))
#else
#define ACF_HIP_API
#endif

enum {
    ACF_HIP_OK = 0,
    ACF_HIP_E_INVALID = 1,     /* bad argument / precondition (reference: CV_Assert) */
    ACF_HIP_E_UNSUPPORTED = 2, /* option the HIP path does not implement */
    ACF_HIP_E_NOMODEL = 3,     /* set_model not called */
    ACF_HIP_E_NOPLAN = 4,      /* plan not called */
    ACF_HIP_E_HIP = 5,         /* HIP runtime error, see acf_hip_last_error */
    ACF_HIP_E_NODEVICE = 6,    /* no gfx950 device visible */
    ACF_HIP_E_CAPACITY = 7     /* more detections than the planned capacity */
};

/* rgbConvert flags (rgbConvert.cpp:106-128) */
enum {
    ACF_HIP_CS_GRAY = 0,
    ACF_HIP_CS_RGB = 1,
    ACF_HIP_CS_LUV = 2,
    ACF_HIP_CS_HSV = 3,
    ACF_HIP_CS_ORIG = 4
};

/*
 * Model + options: the fields of Detector::Classifier (ACF.h:292-310) and of
 * the Options tree (ACF.h:68-275) that the hot path reads.  Tree arrays are
 * row-major [nTrees][nTreeNodes] (the layout after the load-time transpose,
 * ACFIO.cpp:61-67) and are copied by acf_hip_set_model.
 */
typedef struct acf_hip_params
{
    /* Classifier */
    int32_t nTrees;
    int32_t nTreeNodes;
    int32_t treeDepth; /* 1..8 fixed depth, 0 = walk child[] (acfDetect1.cpp:100-166) */
    const uint32_t* fids;
    const float* thrs;
    const float* hs;
    const uint32_t* child; /* may be NULL when treeDepth > 0 */

    /* Options: modelDs, modelDsPad, stride, cascThr (ACF.h:204-211) */
    int32_t modelDs_h, modelDs_w;
    int32_t modelDsPad_h, modelDsPad_w;
    int32_t stride;
    double cascThr;

    /* Options::Pyramid (ACF.h:97-202) */
    int32_t nPerOct, nOctUp, nApprox;
    int32_t nLambdas; /* 3 (colour, gradMag, gradHist), or 0: estimated from every image (chnsPyramid.cpp:341-374) */
    double lambdas[3];
    int32_t pad_h, pad_w;
    int32_t minDs_h, minDs_w;
    double smooth; /* final channel smoothing radius (chnsPyramid.cpp:399-407) */

    /* Options::Pyramid::Chns (ACF.h:99-182) */
    int32_t shrink;
    int32_t colorEnabled;
    double colorSmooth;
    int32_t colorSpace; /* ACF_HIP_CS_* */
    int32_t gradMagEnabled;
    int32_t colorChn;
    int32_t normRad;
    double normConst;
    int32_t full;
    int32_t gradHistEnabled;
    int32_t binSize; /* 0 = shrink */
    int32_t nOrients;
    int32_t softBin;
    int32_t isLuv; /* Detector::setIsLuv (ACF.h:560-567) */

    /* LDCF (BASELINE cfg 5: "k 5x5 filters per channel").  NO reference counterpart ("has not yet been added",
     * README.rst:8): the definition is the upstream toolbox's acfDetectImg — for every level,
     *   C(:,:,j) = conv2(chns(:,:,mod(j-1,nChns)+1), filters(:,:,j), 'same'),  j = 1 .. nChns*k   (zero-padded true convolution)
     *   level    = imResample(C, .5)                                                 (size round(.5 * size))
     * and the cascade then runs with shrink*2 over nChns*k channels (fids index nChns*k*(modelDsPad/(2*shrink))^2 cells).
     * ldcfFilters: [k][nChns][5][5] floats in the MATLAB memory order of fs(:,:,c,f): tap (dy, dx) at dy + 5*dx.
     * The summation order of the 25 taps (dx ascending, then dy ascending, one chain of
     * fused multiply-adds acc = fmaf(v, w, acc) from 0: a single rounding per tap) is this repo's own.
     * ldcfK == 0 or ldcfFilters == NULL: off. */
    int32_t ldcfK;
    const float* ldcfFilters;
} acf_hip_params;

/* One detection in upright image coordinates (ACF.cpp:302-312, Detection ACF.h:510-525). */
typedef struct acf_hip_detection
{
    int32_t x, y, w, h;
    float score;
    int32_t scale; /* pyramid level that produced it */
} acf_hip_detection;

/* bbNms (types max / maxg; ms and cover are pass-through stubs in the reference, bbNms.cpp:100-108) and
 * ObjectDetector::prune, run on the device on every frame's detections (bbNms.cpp:111-192,229-304;
 * ObjectDetector.cpp:28-44).  Equal scores keep their detection order (the reference's std::sort leaves it open). */
typedef struct acf_hip_nms_params
{
    int32_t type;       /* 0 none, 1 max, 2 maxg (greedy) */
    int32_t ovrDnmUnion;/* 1: overlap / union (ovrDnm "union"), 0: overlap / smaller area ("min") */
    double overlap;     /* suppress when the ratio exceeds this */
    double thr;         /* scores below thr are dropped first (reference default: -DBL_MAX) */
    int32_t prune;      /* 1: apply ObjectDetector::prune to the survivors */
    int32_t maxCount;   /* m_maxDetectionCount */
    double pruneRatio;  /* m_detectionScorePruneRatio */
} acf_hip_nms_params;

#define ACF_HIP_NMS_CAP 2048 /* detections per frame the device NMS takes; more is ACF_HIP_E_CAPACITY */

/* One cascade hit before box mapping: DetectionSink::add({c,r},h) (acfDetect1.cpp:39-47,92-95). */
typedef struct acf_hip_hit
{
    int32_t scale, c, r;
    float score;
} acf_hip_hit;

/* Geometry of one pyramid level as planned by acf_hip_plan. */
typedef struct acf_hip_level
{
    double scale;            /* Pyramid::scales (ACF.h:374) */
    double scalehw_h;        /* Pyramid::scaleshw, image-height axis */
    double scalehw_w;        /* Pyramid::scaleshw, image-width axis */
    int32_t isReal;          /* computed exactly (chnsPyramid.cpp:274-277) */
    int32_t realIndex;       /* level it is approximated from (itself if real) */
    int32_t hC, wC;          /* channel plane size before padding (cells) */
    int32_t hP, wP;          /* plane size after BORDER_REFLECT padding = what the cascade sees */
    int32_t nWinR, nWinC;    /* window grid height1, width1 (acfDetect1.cpp:258-259) */
    int64_t offset;          /* float offset of this level inside one frame's fused pyramid */
} acf_hip_level;

typedef struct acf_hip_ctx acf_hip_ctx;

/* ---- lifecycle ------------------------------------------------------- */

/* Bind a context to `device` and to `stream` (a hipStream_t, or NULL for a
 * stream owned by the context).  Replaces constructing an acf::Detector
 * (ACF.h:59-66); good() == (return value == ACF_HIP_OK). */
ACF_HIP_API int acf_hip_create(int device, void* stream, acf_hip_ctx** out);
/* Number of gfx950 devices visible to the process (0 and ACF_HIP_E_NODEVICE when there is none): what a multi-device host
 * (acf::HipDetectorPool, one context per device) iterates over. */
ACF_HIP_API int acf_hip_device_count(int* count);
ACF_HIP_API int acf_hip_destroy(acf_hip_ctx* ctx);
ACF_HIP_API int acf_hip_abi_version(void);
/* Last error text for this context (thread-compatible, like one Detector per thread). */
ACF_HIP_API const char* acf_hip_last_error(const acf_hip_ctx* ctx);

/* Runtime knobs, the counterpart of the reference's setters (ACF.h:495-595).
 * Keys: "taps" (0/1: keep per-stage intermediates readable through
 * acf_hip_read_tap, the role of setLogger's MatLoggerType tap,
 * chnsCompute.cpp:241-250,285-300; costs one extra full-resolution write; set
 * before acf_hip_plan), "profile" (0/1: record HIP events around every kernel,
 * read with acf_hip_profile_get), "scale_streams" (1, default: the real scales of a
 * batch run concurrently on streams of the context — lowest latency and best
 * throughput for ONE context; 0: everything on the context's stream in order, for
 * applications that run several contexts side by side on one GPU), "rank_cells" (1,
 * default: the cascade of a depth-2 model reads 16-bit threshold-rank cells — the
 * comparison `chns[cid] < thrs[node]` of acfDetect1.cpp:102-104 decided on
 * rank(cell) < rank index of the threshold, the same decision for every cell and
 * node, hence identical hits and scores — instead of the float pyramid; 0: floats),
 * "keep_pyramid" (1, default; 0: the caller wants detections only — acf_hip_run /
 * acf_hip_pyramid + acf_hip_detect — and when the levels leave as rank cells the float
 * pyramid is not written at all: acf_hip_read_level then fails with ACF_HIP_E_INVALID),
 * "cascade_turns" (0, default; for applications that run several contexts on one device,
 * the way the reference runs one Detector per thread: bit 0 = a context's cascade tile
 * kernel waits for the tile kernel submitted before it on that device, whichever
 * context's, bit 2 = the level kernel takes its turns in the same chain (bit 1: in a
 * chain of its own) — these kernels are bound by VALU and LDS, and two of them side by
 * side displace each other where each could run beside another context's memory-bound
 * pyramid kernels; three contexts, cfg 2: +4 % frames/s with 5),
 * "tile_persist" (1, default: the cascade's tile kernel runs as persistent workgroups that
 * draw their tiles from a counter — best for a context that has the device to itself;
 * 0: one workgroup per tile, which lets other contexts' kernels find free LDS between
 * tiles — what an application with several contexts per device wants; n > 1: n workgroups),
 * "fused_grad" (1, default: gradMag is computed inside the gradient plane's smoothing
 * chain — the smoothed plane then makes no round trip through memory — for planes of
 * >= 2^20 pixels in batches of >= 16 frames, where that pays; 2: wherever that kernel
 * applies; 0: always as its own kernel),
 * "fused_tri" (1, default: convTri's x pass over M — chnsCompute.cpp:283's convTri(M, S,
 * normRad) — rides on that chain as well, so M is written once and not read back by a
 * kernel of its own, in batches of >= 64 frames of a context with "shared_device" set; 2:
 * wherever "fused_grad"'s kernel runs; 0: never.  The gradient plane is then one uncut
 * chain of column steps per frame),
 * "shared_device" (0, default; 1: this context runs beside other contexts on the same
 * device — acf::HipDetectorPool and the Python DetectorPool set it.  Kernel forms are then
 * chosen for the least total work instead of the shortest time alone: the smoothing
 * chains of batches >= 64 frames stay uncut — no warm-up columns, no verify / repair
 * launch — and "fused_tri" applies; 3 contexts x 96 frames at 1080p: +4.3 % frames/s,
 * one context alone: -7 %),
 * "graph" (0, default; 1: acf_hip_run captures its own launches into a HIP graph the
 * second time it is called with the same frames pointer and batch size, and replays it
 * from then on — one host call instead of ~45 launches, for callers that feed one
 * frame at a time from a fixed buffer; not with "profile", taps, image-specific
 * lambdas, "cascade_turns" or sub-batch contexts ("streams" > 1), where the call runs
 * plainly; any set_* / plan call drops the graph),
 * "level_segments" (0, default = auto: batches of at most 8 frames cut every level's column
 * chain into speculative segments that are verified bit for bit on the device and repaired by a
 * whole-chain launch where a hand-over differs — a single frame's latency 1.05 -> 0.78 ms;
 * bigger batches run one chain per plane; 1: never; n > 1: n segments whatever the batch),
 * "level_warm" (32, default: warm-up columns of a level segment, a positive multiple of 4),
 * "smooth_segments" / "smooth_warm" (the same for the image smoothing: 0 = as many segments as
 * fill the device, 64 warm-up columns), "count_repairs" (1: acf_hip_get_repairs counts the
 * planes the repair launches recomputed; synchronises, measurements only),
 * "smooth_force_redo" (tests: 1 = every plane with more than one segment is marked for the
 * repair launch whatever the verification found),
 * "cascade_tiles", "fused_levels", "fused_smooth", "streams" (kernel-form A/B
 * switches; all forms give identical results),
 * "arith" (0, default: the three sites where the reference's SSE kernels use _mm_rsqrt_ps /
 * _mm_rcp_ps — gradMag, gradMagNorm, rgb2luv_sse: toolbox/gradientMex.cpp:209-219,266,
 * toolbox/rgbConvertMex.cpp:161, toolbox/sse.hpp:185-192 — compute 1/sqrt and 1/x exactly;
 * 1: they return the bits of one x86 CPU's instructions, from the tables installed with
 * acf_hip_set_x86_tables — the pyramid and the detections are then what the reference's own
 * compiled kernels give on that CPU, bit for bit.  Every kernel form of the default path has
 * this arithmetic too (the tables ride in LDS beside the acos table): ~94 % of the default
 * tier's frames/s).
 *
 * Capacity: `max_hits` of acf_hip_plan bounds the hits kept per frame.  With stride < shrink
 * (the cascade then runs once per distinct cell offset and k_expand_hits writes every window
 * of a surviving offset) a frame whose EXPANDED total exceeds max_hits reports the overflow
 * in its count as always, but WHICH of its hits are retained is unspecified (slots are
 * reserved in thread order; a q x q group may be partly written).  "keep_pyramid" = 0 has no
 * effect for LDCF models, for stride < shrink and for tree depths 1, 3 and 4 on rank cells:
 * those paths read the float pyramid (their overflow queues do), so it is always written. */
ACF_HIP_API int acf_hip_set_option(acf_hip_ctx* ctx, const char* key, int value);

/* The reference's arithmetic at its three approximate sites (option "arith" = 1).  _mm_rcp_ps and _mm_rsqrt_ps
 * (toolbox/sse.hpp:185-192) are 12-bit approximations whose bits belong to the CPU.  On the CPUs probed (an Intel Xeon and an AMD
 * EPYC, each checked for all 2^32 inputs) they are functions of the input's sign, exponent (its parity for rsqrt) and top 12
 * mantissa bits:
 *   rcp4096[m >> 11]                      = bits of _mm_rcp_ps(x)   for x = 1.m in [1, 2)        (4096 entries)
 *   rsqrt8192[(odd << 12) | (m >> 11)]    = bits of _mm_rsqrt_ps(x) for x in [1, 2) (odd = 0) and [2, 4) (odd = 1)
 * and every other input follows by exponent arithmetic (zero / subnormal -> inf, inf -> 0, results below the normal range
 * -> 0, NaN quieted, rsqrt of a negative -> 0xffc00000).  A host that wants "what the reference gives HERE" fills the tables
 * from its own CPU (acf::HipDetector::setReferenceArithmetic does, acf_amd/host/HipDetector.h); tests install the build
 * host's (tests/golden/x86_rcp_rsqrt.npz) or probe the host they run on.  The tables are copied to the device; the call may be
 * repeated.  Sub-batch contexts ("streams" > 1) share their parent's tables. */
ACF_HIP_API int acf_hip_set_x86_tables(acf_hip_ctx* ctx, const uint32_t* rcp4096, const uint32_t* rsqrt8192);
/* Self-check of the device's table functions: position-mixed 64-bit digests of rcp (digest[0]) and rsqrt (digest[1]) over the
 * bit patterns first + i * stride, i < count — the sums the CPU oracle's acfo_x86_digest forms from the same tables, so the
 * two implementations are compared for every input without moving 2^32 results — and digest[2] = the number of those patterns
 * (taken as gradMag's M2: the negative non-NaN ones skipped) for which the column kernels' one-read form of
 * m = min(rsqrt(M2), 1e10), M = rcp(m) differs from the two table functions applied in turn: must be 0. */
ACF_HIP_API int acf_hip_selftest_x86(acf_hip_ctx* ctx, uint32_t first_bits, uint64_t count, uint32_t stride, uint64_t digest[3]);

/* Detector::getScales (static, chnsPyramid.cpp:461-529): host only, no context.
 * Writes up to `cap` scales and returns the total count in *n. */
ACF_HIP_API int acf_hip_get_scales(int nPerOct, int nOctUp, int minDs_h, int minDs_w, int shrink, int h, int w,
    double* scales, double* scaleshw_h, double* scaleshw_w, int cap, int* n);

/* The level geometry acf_hip_plan would produce for these options and frame
 * size, computed on the host without touching a device (what chnsPyramid's
 * bookkeeping, chnsPyramid.cpp:270-292, and createDetector's window grid,
 * acfDetect1.cpp:258-259, yield). */
ACF_HIP_API int acf_hip_plan_levels(const acf_hip_params* p, int h, int w, int d, acf_hip_level* out, int cap, int* nScales, int* nChns);

/* Upload classifier + options.  Replaces Detector::deserialize*() filling
 * `clf` and `opts` (ACF.h:277,312) and acfModify's effects (acfModify.cpp:139-143),
 * which the caller applies to the struct before the call. */
ACF_HIP_API int acf_hip_set_model(acf_hip_ctx* ctx, const acf_hip_params* p);

/* Plan for frames of h x w with `d` input planes (1 or 3; or 5 = three image planes followed by the gradient magnitude and
 * orientation that come WITH the image — the GL pipeline's LUVMO frames, chnsPyramid.cpp:248-255: M and O then replace
 * gradientMag at the first real scale, :318-322, which must be the image's own size: nOctUp = 0) and up to
 * `max_batch` frames per call and `max_hits` hits per frame: runs getScales
 * (chnsPyramid.cpp:461-529) and the real/approximate split
 * (chnsPyramid.cpp:272-292), builds resampling tables and allocates every
 * device buffer.  Nothing is allocated after this call. */
ACF_HIP_API int acf_hip_plan(acf_hip_ctx* ctx, int h, int w, int d, int max_batch, int max_hits);
ACF_HIP_API int acf_hip_num_levels(const acf_hip_ctx* ctx, int* nScales, int* nChns);
ACF_HIP_API int acf_hip_get_levels(const acf_hip_ctx* ctx, acf_hip_level* out, int cap);
/* Geometry of the LDCF levels the cascade reads when the model has ldcfK > 0 (hP, wP = round(.5 * size), window grid at
 * shrink*2, offsets inside one frame's LDCF pyramid); same count as acf_hip_get_levels.  ACF_HIP_E_INVALID without LDCF. */
ACF_HIP_API int acf_hip_get_ldcf_levels(const acf_hip_ctx* ctx, acf_hip_level* out, int cap);
/* Floats in one frame's fused pyramid (sum over levels of nChns*wP*hP). */
ACF_HIP_API int acf_hip_pyramid_floats(const acf_hip_ctx* ctx, int64_t* n);
/* The lambdas frame `frame` of the last acf_hip_pyramid / acf_hip_run was approximated with: the model's, or — for a
 * model without lambdas — the ones estimated from that image (Detector::Pyramid::lambdas, chnsPyramid.cpp:341-374). */
ACF_HIP_API int acf_hip_get_lambdas(acf_hip_ctx* ctx, int frame, double out[3]);

/* ---- the hot path ---------------------------------------------------- */

/* Detector::chnsPyramid for a batch (chnsPyramid.cpp:160-456).  `frames_dev`:
 * n_frames x d planes of float[w][h] already on the device.  Asynchronous on
 * the context's stream.
 * Values: the reference feeds images in [0, 1] (ACF.cpp:119).  Bit parity with it is checked for every finite input whose
 * squared gradients stay finite (|values| up to ~1e18: flat regions, denormal-sized and very large gradients included,
 * tests/test_gpu_arith.py); frames holding NaN / inf, or values whose squares overflow f32, are outside the contract (the
 * reference itself turns those into inf / NaN channels). */
ACF_HIP_API int acf_hip_pyramid(acf_hip_ctx* ctx, const float* frames_dev, int n_frames);

/* Detector::operator()(const Pyramid&) without NMS (ACF.cpp:268-367) on the
 * pyramid of the last acf_hip_pyramid call: acfDetect1 on every level
 * (acfDetect1.cpp:309-335) + box mapping.  Asynchronous. */
ACF_HIP_API int acf_hip_detect(acf_hip_ctx* ctx);

/* acf_hip_pyramid + acf_hip_detect: Detector::operator()(const MatP&) (ACF.cpp:246-265). */
ACF_HIP_API int acf_hip_run(acf_hip_ctx* ctx, const float* frames_dev, int n_frames);

/* Same with frames in host memory (H2D copy included). */
ACF_HIP_API int acf_hip_run_host(acf_hip_ctx* ctx, const float* frames_host, int n_frames);

/* Pixel layouts of packed 8-bit frames: the cv::Mat inputs of the image entry
 * Detector::operator()(const cv::Mat&) (CV_8UC3 RGB, ACF.cpp:135-141) and the
 * BGR/BGRA/grey video frames the apps convert before calling it (acf.cpp:117-148,
 * GPUDetectionPipeline.cpp:250-266).  Alpha is ignored. */
enum {
    ACF_HIP_PIX_RGB = 0,
    ACF_HIP_PIX_BGR = 1,
    ACF_HIP_PIX_RGBA = 2,
    ACF_HIP_PIX_BGRA = 3,
    ACF_HIP_PIX_GRAY = 4
};

/* The image entry for 8-bit input: cvt8UC3To32FC3 = convertTo(CV_32F, 1/255)
 * (ACF.cpp:114-119), I.t() (ACF.cpp:137) and the MatP plane split
 * (MatP.cpp:51-73) in one kernel, fused with the colour conversion of
 * chnsPyramid.cpp:230-263 when the model asks for one; then as acf_hip_pyramid.
 * `frames_dev`: n_frames upright images of h rows x w pixels, `row_stride_bytes`
 * between rows (0 = tightly packed), frames back to back (h * stride bytes
 * apart).  The plan's `d` must be 3 for the colour layouts and 1 for GRAY. */
ACF_HIP_API int acf_hip_pyramid_u8(acf_hip_ctx* ctx, const uint8_t* frames_dev, int n_frames, int pix, int row_stride_bytes);
ACF_HIP_API int acf_hip_run_u8(acf_hip_ctx* ctx, const uint8_t* frames_dev, int n_frames, int pix, int row_stride_bytes);

/* The apps' resize to a minimum object width (src/app/acf/acf.cpp:117-148 `Resizer`,
 * GPUDetectionPipeline.cpp:250-266 computeDetectionWidth): scale = float(winSize.width) /
 * float(minWidth); cv::resize(image, reduced, {}, scale, scale, scale < 1 ? INTER_AREA :
 * INTER_LINEAR) on the packed 8-bit frame, the detector on `reduced`, the boxes back by
 * cv::Rect2f(o) * (1.f / scale).  cv::resize is OpenCV's (not part of the reference tree):
 * the CV_8U arithmetic of its imgproc/resize.cpp is restated (DESIGN.md 6b says what
 * exactly); PARITY UNPINNED.
 *  acf_hip_resize_dims       the reduced size cv::resize produces: cvRound(rows * scale) x cvRound(cols * scale)
 *  acf_hip_set_input_resize  after acf_hip_plan FOR THE REDUCED SIZE: from now on the 8-bit entries
 *                            (acf_hip_pyramid_u8 / run_u8 / stream_*) take frames of src_rows x src_cols and
 *                            reduce them on the device first (k_resize_u8); src_rows = 0 switches it off; a new
 *                            plan switches it off
 *  acf_hip_op_resize_u8      the resize alone, host buffers (dst: dst_rows x dst_cols x cpp, tight) */
ACF_HIP_API int acf_hip_resize_dims(int rows, int cols, double scale, int* out_rows, int* out_cols);
ACF_HIP_API int acf_hip_set_input_resize(acf_hip_ctx* ctx, int src_rows, int src_cols, double scale);
ACF_HIP_API int acf_hip_op_resize_u8(acf_hip_ctx* ctx, const uint8_t* src_host, int rows, int cols, int cpp, int row_stride_bytes, double scale,
    uint8_t* dst_host, int dst_rows, int dst_cols);

/* ---- streaming front end ---------------------------------------------
 * The overlap of transfer and compute that GPUDetectionPipeline::runFast gets
 * from its two-frame texture FIFO (GPUDetectionPipeline.cpp:357-437), expressed
 * with HIP streams and events: batch k+1 is copied host->device on a copy
 * stream while batch k runs on the context's stream; results come back as the
 * fixed-capacity records of acf_hip_export_detections in pinned host memory.
 *
 *   acf_hip_stream_open(ctx, pix, stride, cap, depth)   depth = batches in flight (2..8)
 *   acf_hip_stream_submit(ctx, frames_host, n, &ticket)  returns at once (frames_host pinned: see acf_hip_host_alloc)
 *   acf_hip_stream_collect(ctx, ticket, &records, &n)    waits for that batch only
 *
 * Tickets must be collected in order; `records` stays valid until `depth` more
 * batches have been submitted.  `frames_host` must stay untouched until its
 * ticket is collected. */
ACF_HIP_API int acf_hip_stream_open(acf_hip_ctx* ctx, int pix, int row_stride_bytes, int cap, int depth);
ACF_HIP_API int acf_hip_stream_submit(acf_hip_ctx* ctx, const uint8_t* frames_host, int n_frames, int* ticket);
ACF_HIP_API int acf_hip_stream_collect(acf_hip_ctx* ctx, int ticket, const int32_t** records, int* n_frames);
ACF_HIP_API int acf_hip_stream_close(acf_hip_ctx* ctx);
/* Page-locked host memory for acf_hip_stream_submit / acf_hip_run_host callers that do not link HIP. */
ACF_HIP_API int acf_hip_host_alloc(size_t bytes, void** out);
ACF_HIP_API int acf_hip_host_free(void* p);

/* Wait for the stream, then return frame `frame`'s detections in the
 * reference's order (level ascending, then c, then r; ACF.cpp:326-329,
 * acfDetect1.cpp:86-96).  `*count` is the true number even if > cap. */
/* params != NULL: from the next acf_hip_detect / acf_hip_run* on, every frame's detections go through bbNms (+ prune)
 * on the device, and acf_hip_get_detections / acf_hip_export_detections / the stream records return the survivors, in
 * score order (Detector::operator(), ACF.cpp:332-353).  NULL: back to the raw list (scale, column, row order). */
ACF_HIP_API int acf_hip_set_nms(acf_hip_ctx* ctx, const acf_hip_nms_params* params);
/* bbNms + prune of one host list: boxes [n][4] = {x, y, w, h}, f64 scores; keep_idx receives the indices of the survivors
 * in output order (capacity n), *count their number.  n <= ACF_HIP_NMS_CAP. */
ACF_HIP_API int acf_hip_op_nms(acf_hip_ctx* ctx, const int32_t* boxes, const double* scores, int n, const acf_hip_nms_params* params,
    int32_t* keep_idx, int* count);
ACF_HIP_API int acf_hip_get_detections(acf_hip_ctx* ctx, int frame, acf_hip_detection* out, int cap, int* count);
/* The cascade's hits behind the detections: hit i is the window that produced detection i of acf_hip_get_detections (with
 * the device NMS on: the survivors' windows, in the survivors' order; *count = their number). */
ACF_HIP_API int acf_hip_get_hits(acf_hip_ctx* ctx, int frame, acf_hip_hit* out, int cap, int* count);
/* The list acfDetect1 produced (ACF.cpp:302-329: scale, column, row order) whether or not the device NMS is on.  The device
 * NMS takes at most ACF_HIP_NMS_CAP detections per frame; beyond that acf_hip_get_detections returns ACF_HIP_E_CAPACITY for the
 * frame (record count -1) and the caller suppresses this list on the host (acf::HipDetector does; bbNms.cpp has no limit). */
ACF_HIP_API int acf_hip_get_raw_detections(acf_hip_ctx* ctx, int frame, acf_hip_detection* out, int cap, int* count);

/* Device-side export for the multi-GPU gather: writes, for every frame of the
 * last batch, a fixed-capacity record [count, then cap x {x,y,w,h,score bits,scale}]
 * of int32 into `dst_dev` (n_frames * (1 + 6*cap) int32), sorted as above. */
ACF_HIP_API int acf_hip_export_detections(acf_hip_ctx* ctx, int32_t* dst_dev, int cap);

ACF_HIP_API int acf_hip_synchronize(acf_hip_ctx* ctx);
/* With option "count_repairs" = 1 (measurements: it synchronises after every verification): out = {image planes whose
 * smoothing segments were checked, of those recomputed as one chain, level planes checked, recomputed} since the context
 * was created — how often the speculative column segments (options smooth_segments / smooth_warm / level_warm) miss. */
ACF_HIP_API int acf_hip_get_repairs(acf_hip_ctx* ctx, int64_t out[4]);

/* Per-kernel timing with HIP events recorded on the context's stream (option
 * "profile" = 1): the counterpart of the reference's ScopeTimeLogger stage
 * timers (GPUDetectionPipeline.cpp:366,415,485).  Synchronises, then returns
 * up to `cap` kernel names (static strings), their accumulated milliseconds and
 * launch counts since the last call, and resets the accumulation. */
ACF_HIP_API int acf_hip_profile_get(acf_hip_ctx* ctx, int* n, const char** names, float* ms, int* launches, int cap);

/* ---- parity taps (tests only; not on the timed path) ------------------ */

/* Copy level `level` of frame `frame`'s fused pyramid ([nChns][wP][hP]) to host. */
ACF_HIP_API int acf_hip_read_level(acf_hip_ctx* ctx, int frame, int level, float* host_out);
/* The same level as the 16-bit threshold-rank cells the cascade read ([nChns][wP][hP], after acf_hip_detect / acf_hip_run
 * with "rank_cells" on and a model that admits them; ACF_HIP_E_UNSUPPORTED otherwise). */
ACF_HIP_API int acf_hip_read_rank_level(acf_hip_ctx* ctx, int frame, int level, uint16_t* host_out);
/* Host only, no context: the rank cells of `n` values of channel `chn` and the rank indices of every node threshold, as
 * acf_hip_plan builds them for this model (`nChns` channels).  rank(v) = number of distinct node thresholds t of the
 * channel with t <= v; node q's index is k + 1 where t_k is its threshold, so that v < thrs[q]  <=>  rank(v) < index[q].
 * cells_out[n]; thr_index_out[nTrees * nTreeNodes] (0 for leaves); either may be NULL.  info[4] = {usable, shift, buckets,
 * distinct thresholds} of the channel.  ACF_HIP_E_UNSUPPORTED when the model's thresholds do not admit the table. */
ACF_HIP_API int acf_hip_rank_cells_host(const acf_hip_params* params, int nChns, int chn, const float* v, int n, uint16_t* cells_out,
    uint32_t* thr_index_out, int32_t* info);

enum {
    ACF_HIP_TAP_IMAGE = 0,    /* resampled image at a real scale, before smoothing: d planes */
    ACF_HIP_TAP_SMOOTHED = 1, /* after convTri (logger tags L,U,V chnsCompute.cpp:241-250) */
    ACF_HIP_TAP_M = 2,        /* gradMag before normalisation (tag M, gradientMag.cpp:112-117) */
    ACF_HIP_TAP_O = 3,        /* orientation (tag O) */
    ACF_HIP_TAP_S = 4,        /* convTri(M, normRad) */
    ACF_HIP_TAP_MNORM = 5,    /* tag Mnorm */
    ACF_HIP_TAP_CHNS = 6,     /* unsmoothed, unpadded channels of a level: nChns planes [wC][hC] */
    ACF_HIP_TAP_LDCF = 7      /* LDCF level the cascade reads: nChns*k planes [round(wP/2)][round(hP/2)] (after acf_hip_detect) */
};
/* `index`: real-scale ordinal for taps 0-5, level for ACF_HIP_TAP_CHNS. */
ACF_HIP_API int acf_hip_read_tap(acf_hip_ctx* ctx, int frame, int tap, int index, float* host_out, int64_t cap_floats);

/* ---- single operators on host planes ---------------------------------
 * The reference exposes these as static members of acf::Detector
 * (ACF.h:441-493) and as free functions; each runs the same HIP kernel the
 * pyramid uses, on one plane set, synchronously. */

/* Detector::rgbConvert (rgbConvert.cpp:101-170) flag = ACF_HIP_CS_LUV or GRAY. */
ACF_HIP_API int acf_hip_op_rgb_convert(acf_hip_ctx* ctx, const float* in, float* out, int h, int w, int flag);
/* Detector::convTri (convTri.cpp:204-253): r in (0,1] -> convTri1 with the
 * reference's in-place aliasing semantics when `aliased` != 0; r > 1 -> convTri. */
ACF_HIP_API int acf_hip_op_conv_tri(acf_hip_ctx* ctx, const float* in, float* out, int h, int w, int d, double r, int aliased);
/* Detector::gradientMag (gradientMag.cpp:102-135). S_out may be NULL. */
ACF_HIP_API int acf_hip_op_gradient_mag(acf_hip_ctx* ctx, const float* in, float* M, float* O, float* S_out,
    int h, int w, int normRad, double normConst, int full);
/* Self-check of the gradMag kernels' fast form of m = min(1/sqrt(m2), 1e10), M = 1/m (one v_rsq_f32 + FMA refinements)
 * against the IEEE sqrt and divisions it stands for (gradientMex.cpp:209-219 with exact arithmetic), on the device, for
 * every float bit pattern first_bits .. last_bits taken as m2: *mismatches = how many differ in either result,
 * *first_bad_bits = the smallest such pattern.  0 .. 0x7f7fffff (every finite m2 >= 0) must give 0. */
ACF_HIP_API int acf_hip_selftest_gradmag(acf_hip_ctx* ctx, uint32_t first_bits, uint32_t last_bits, uint64_t* mismatches, uint32_t* first_bad_bits);
/* Detector::chnsCompute (ACF.h:342-349, chnsCompute.cpp:146-338; addChn :340-370): the channels of ONE image at its own scale,
 * without a pyramid plan — crop to a multiple of shrink (:203-217), rgbConvert (:235; skipped for colorSpace orig / rgb and for
 * isLuv), convTri(I, I, pColor.smooth, 1) in place (:239), gradientMag with its normalisation (:263-283), gradientHist (:310-331),
 * every enabled type reduced by `shrink` (addChn's imResample) and concatenated: colour, magnitude, histogram.
 *   p     the Options::Pyramid::Chns fields of acf_hip_params are read (shrink .. isLuv; classifier and pyramid fields ignored);
 *         NULL: the context's model (acf_hip_set_model)
 *   in    HOST planes [d][w][h], d = 1 or 3 — or 5: three image planes followed by the gradient magnitude and orientation that came
 *         with the image (chnsCompute.cpp:219-226: they replace gradientMag and its normalisation) —, the transposed planar layout
 *   out   HOST buffer of `cap` floats receiving [nChns][w / shrink][h / shrink] (cropped sizes); NULL: only the sizes are reported
 * *nChns, *hC, *wC (each may be NULL) receive the channel count and the cell-plane size.  Honours option "arith". */
ACF_HIP_API int acf_hip_chns_compute(acf_hip_ctx* ctx, const acf_hip_params* p, const float* in, int h, int w, int d, float* out, int64_t cap,
    int* nChns, int* hC, int* wC);
/* Detector::gradientHist (gradientHist.cpp:92-115) -> gradHist (gradientMex.cpp:375-509): soft_bin even — >= 0 the magnitude
 * is shared between the two nearest orientation bins, < 0 the nearest bin takes it; no spatial interpolation.  Odd soft_bin
 * (trilinear: HOG / FHOG features) is ACF_HIP_E_UNSUPPORTED. */
ACF_HIP_API int acf_hip_op_gradient_hist(acf_hip_ctx* ctx, const float* M, const float* O, float* H,
    int h, int w, int bin, int nOrients, int soft_bin, int full);
/* imResample (imResampleMex.cpp:385-420). */
ACF_HIP_API int acf_hip_op_im_resample(acf_hip_ctx* ctx, const float* in, float* out, int ha, int wa, int hb, int wb, int d, double nrm);
/* Detector::acfDetect1 on one host channel buffer [nChns][wP][hP] (acfDetect1.cpp:309-335). */
ACF_HIP_API int acf_hip_op_acf_detect1(acf_hip_ctx* ctx, const float* chns, int hP, int wP, int nChns,
    acf_hip_hit* out, int cap, int* count);

/* The uint8_t body of acfDetect1 (ParallelDetectionBody<uint8_t,...>, acfDetect1.cpp:157-166,187-192) on one host
 * channel buffer of bytes [nChns][wP][hP], the format the reference's GL backend feeds the cascade (GPUACF.cpp:790).
 * `thrsU8`: Classifier::thrsU8 ([nTrees][nTreeNodes] bytes), or NULL to derive it from the model's thrs as the
 * loader does (ACFIOArchive.h:96-99, acf_hip_thrs_u8). */
ACF_HIP_API int acf_hip_op_acf_detect1_u8(acf_hip_ctx* ctx, const uint8_t* chns, int hP, int wP, int nChns, const uint8_t* thrsU8,
    acf_hip_hit* out, int cap, int* count);
/* Detector::evaluate(const MatP&, shrink, modelDsPad, stride) (ACF.h:543-544, acfDetect1.cpp:337-342): score of the single
 * window at (0,0) of one host channel buffer [nChns][wP][hP]; trees are added until the score is <= cascThr (the reference
 * passes 0 here) and the score reached is returned, whether or not the window would be a detection. */
ACF_HIP_API int acf_hip_op_evaluate(acf_hip_ctx* ctx, const float* chns, int hP, int wP, int nChns, double cascThr, float* score);
/* thrs.convertTo(thrsU8, CV_8UC1, 255.0f) (ACFIOArchive.h:96-99): host only, no context. */
ACF_HIP_API int acf_hip_thrs_u8(const float* thrs, int n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ACF_HIP_H */
