// HipDetector.h — host-side C++ mirror of the reference's acf::Detector for the
// chnsPyramid + acfDetect hot path, backed by libacf_hip.so (MI355X / gfx950).
//
// Same method names, argument meaning and error behaviour as acf::Detector
// (reference src/lib/acf/acf/ACF.h:50-624, ObjectDetector.h:31-48); it plays
// the role acf::GLDetector plays for the GL backend (src/app/acf/GLDetector.h:
// 28-52): build the pyramid on the device, then run the multi-scale search.
// OpenCV is not available in this build image, so the three cv:: value types
// the API traffics in are restated here with the same member names; a
// maintainer wiring this into the reference replaces them by the cv:: ones
// (INTEGRATION.md shows the subclass).
//
// Layout contract (reference MatP.cpp:51-73, ACF.cpp:137): every plane is the
// TRANSPOSED image, float[rows = image width][cols = image height] with
// image-y contiguous.  cv::Size members for modelDs/modelDsPad/pad/minDs hold
// {width = image-height axis, height = image-width axis} (ACFIO.h:168-181);
// this class keeps that convention at its surface and converts to the C
// ABI's upright h/w names internally.
#pragma once

#include "acf_hip_loader.h"

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <iosfwd>
#include <string>
#include <memory>
#include <vector>

namespace acf
{

struct Size
{
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};

struct Size2d
{
    double width = 0, height = 0;
};

struct Rect
{
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
    bool operator==(const Rect& o) const { return x == o.x && y == o.y && width == o.width && height == o.height; }
};

// Planar float image (reference MatP.h:25-190): `channels` planes of rows x cols,
// stored back to back (plane stride rows*cols), cols contiguous.
class MatP
{
public:
    MatP() = default;
    MatP(int rows, int cols, int channels) { create(rows, cols, channels); }
    // non-owning view over caller memory (the reference wraps cv::Mat headers the same way, MatP.cpp:51-73)
    MatP(int rows, int cols, int channels, float* data) : m_rows(rows), m_cols(cols), m_channels(channels), m_ptr(data) {}
    MatP(const MatP& o) { *this = o; }
    MatP& operator=(const MatP& o)
    {
        if (this != &o)
        {
            m_rows = o.m_rows;
            m_cols = o.m_cols;
            m_channels = o.m_channels;
            m_store = o.m_store;
            m_ptr = o.m_store.empty() ? o.m_ptr : m_store.data(); // owning copies re-point; views stay views
        }
        return *this;
    }
    void create(int rows, int cols, int channels)
    {
        m_rows = rows;
        m_cols = cols;
        m_channels = channels;
        m_store.assign(size_t(rows) * cols * channels, 0.f);
        m_ptr = m_store.data();
    }
    int rows() const { return m_rows; }
    int cols() const { return m_cols; }
    int channels() const { return m_channels; }
    bool empty() const { return !m_ptr || !m_rows || !m_cols || !m_channels; }
    Size size() const { return Size(m_cols, m_rows); }
    float* operator[](int c) { return m_ptr + size_t(c) * m_rows * m_cols; }
    const float* operator[](int c) const { return m_ptr + size_t(c) * m_rows * m_cols; }
    float* data() { return m_ptr; }
    const float* data() const { return m_ptr; }
    size_t numel() const { return size_t(m_rows) * m_cols * m_channels; }

private:
    int m_rows = 0, m_cols = 0, m_channels = 0;
    float* m_ptr = nullptr;
    std::vector<float> m_store;
};

// cv::Exception stand-in: precondition failures throw (reference: CV_Assert), the hot path returns 0.
class Exception : public std::runtime_error
{
public:
    Exception(int code_, const std::string& what) : std::runtime_error(what), code(code_) {}
    int code;
};

class HipDetector
{
public:
    using RealVec = std::vector<double>;
    using RectVec = std::vector<Rect>;

    // The fields of the Options tree (ACF.h:68-275) the hot path reads, flattened.
    struct Options
    {
        struct Nms
        {
            std::string type = "maxg"; // acfTrain default pNms; bbNms.cpp:229-304 handles max / maxg / none
            double thr = -1.7976931348623157e308;
            double overlap = 0.65;
            std::string ovrDnm = "min";
        } pNms;
        struct Pyramid
        {
            struct Chns
            {
                int shrink = 4;
                bool isLuv = false; // ACF.h:175: the input planes are LUV already (chnsPyramid.cpp:228 propagates Detector::setIsLuv)
                struct Color { int enabled = 1; double smooth = 1; std::string colorSpace = "luv"; } pColor;
                struct GradMag { int enabled = 1; int colorChn = 0; int normRad = 5; double normConst = 0.005; int full = 0; } pGradMag;
                struct GradHist { int enabled = 1; int binSize = 0; int nOrients = 6; int softBin = 0; } pGradHist;
            } pChns;
            int nPerOct = 8, nOctUp = 0, nApprox = 7;
            std::vector<double> lambdas{ 0.0, 0.1105, 0.1083 };
            Size pad{ 0, 0 };    // {width = image-height axis, height = image-width axis}
            Size minDs{ 16, 16 };
            double smooth = 1;
        } pPyramid;
        Size modelDs{ 16, 16 }, modelDsPad{ 16, 16 };
        // LDCF (toolbox opts.filters, BASELINE cfg 5; not in the reference's Options tree): k filters of 5x5 per channel applied to
        // every level before the cascade, which then runs at shrink*2.  Layout [k][nChns][5][5] = MATLAB memory order of fs(:,:,c,f).
        int ldcfK = 0;
        std::vector<float> ldcfFilters;
        int stride = 4;
        double cascThr = -1;
        double cascCal = 0;
    };

    // Detector::Classifier (ACF.h:292-310): row-major [nTrees][nTreeNodes] (after ACFIO.cpp:61-67's transpose).
    struct Classifier
    {
        int nTrees = 0, nTreeNodes = 0, treeDepth = 0;
        std::vector<uint32_t> fids, child;
        std::vector<float> thrs, hs;
        std::vector<uint8_t> thrsU8; // prescaled thresholds (x255) for uint8_t channels (ACF.h:305); filled on first use if empty
    };

    // Detector::Modify (ACF.h:391-408): the subset acfModify may override (acfModify.cpp:83-152).
    struct Modify
    {
        bool has_nPerOct = false, has_nOctUp = false, has_nApprox = false, has_lambdas = false, has_pad = false,
             has_minDs = false, has_stride = false, has_cascThr = false, has_cascCal = false;
        int nPerOct = 0, nOctUp = 0, nApprox = 0, stride = 0;
        std::vector<double> lambdas;
        Size pad, minDs;
        double cascThr = 0, cascCal = 0;
    };

    // Detector::Pyramid (ACF.h:364-389).  data[i][0] is level i's fused channel
    // buffer [nChns planes][wP rows][hP cols] (ACF.h:653-672 fuseChannels).
    struct Pyramid
    {
        int nTypes = 0, nScales = 0, nChns = 0;
        std::vector<std::vector<MatP>> data;
        std::vector<double> lambdas, scales;
        std::vector<Size2d> scaleshw;
        // ACF.h:377-378 ".rois - [LEVELS x CHANNELS] array for channel access": when rois[i] is not empty, data[i][0] is an
        // ATLAS plane (the GL backend's texture read-back, GPUACF.cpp) whose channels sit side by side: channel z starts
        // (rois[i][1].x - rois[i][0].x) * z elements after the plane's origin, element (c, r) of a channel c * rowStride + r
        // further (computeChannelIndex, acfDetect1.cpp:346-366, GPU_ACF_TRANSPOSE form); rois[i][z].{height, width} = the
        // channel's {columns c, rows r}.  The plane's cols() is its row stride.
        std::vector<std::vector<Rect>> rois;
        void clear()
        {
            rois.clear();
            data.clear();
            lambdas.clear();
            scales.clear();
            scaleshw.clear();
            nScales = 0;
        }
    };

    // Detector::Detection (ACF.h:510-525)
    struct Detection
    {
        Rect roi;
        double score = 0;
    };
    using DetectionVec = std::vector<Detection>;

    // Detector::Channels (ACF.h:326-340): what chnsCompute returns — one MatP per enabled channel type, in the order colour,
    // gradient magnitude, gradient histogram, each [type's channels][w / shrink rows][h / shrink cols]
    struct Channels
    {
        Options::Pyramid::Chns pChns;
        int nTypes = 0;
        std::vector<MatP> data;
        struct Info
        {
            std::string name;
            int nChns = 0;
            std::string padWith;
        };
        std::vector<Info> info;
    };

    // Detector::MatLoggerType (ACF.h:57): called with one plane and a tag "<name>:<cols>x<rows>"
    using MatLoggerType = std::function<int(const MatP&, const std::string&)>;
    // the reference's stream logger is a std::shared_ptr<spdlog::logger> (ACF.h:583-586; spdlog is not in this image): a text sink
    using StreamLoggerType = std::function<void(const std::string&)>;

    Options opts;
    Classifier clf;

    HipDetector() = default; // !good() until a model is supplied
    HipDetector(const Options& o, const Classifier& c, int device = 0);
    // Detector(filename) / Detector(istream, hint) (ACF.h:59-66): "*.cpb" = the reference's cereal files (ModelIO.h)
    explicit HipDetector(const std::string& filename, int device = 0);
    HipDetector(std::istream& is, const std::string& hint = {}, int device = 0);
    ~HipDetector();
    HipDetector(const HipDetector&) = delete;
    HipDetector& operator=(const HipDetector&) = delete;

    bool good() const { return m_good; }
    explicit operator bool() const { return m_good; }
    void setModel(const Options& o, const Classifier& c, int device = 0);

    // ObjectDetector knobs (ObjectDetector.h:37-47, ACF.h:495-595)
    void setDoNonMaximaSuppression(bool flag) { m_doNms = flag; }
    bool getDoNonMaximaSuppression() const { return m_doNms; }
    void setMaxDetectionCount(size_t n) { m_maxDetectionCount = n; }
    void setDetectionScorePruneRatio(double r) { m_detectionScorePruneRatio = r; }
    void setIsLuv(bool flag) { m_isLuv = flag; m_dirty = true; }
    bool getIsLuv() const { return m_isLuv; }
    void setIsTranspose(bool flag) { m_isTranspose = flag; }
    bool getIsTranspose() const { return m_isTranspose; }
    // Detector::setDoParallel (ACF.h:410: cv::parallel_for_ over scales): the real scales of a batch on streams of the context
    // beside each other (default, lowest latency) or all on one stream (several detectors side by side on one GPU)
    void setDoParallel(bool flag);
    // acf_hip_set_option (include/acf_hip.h): backend knobs without a counterpart among the reference's setters
    void setOption(const std::string& key, int value);
    void setIsRowMajor(bool flag) { m_isRowMajor = flag; } // ACF.h:588-595: stored for callers that orient the window size by it
    bool getIsRowMajor() const { return m_isRowMajor; }
    Size getWindowSize() const { return opts.modelDs; }
    // The apps' Resizer (src/app/acf/acf.cpp:117-148; GPUDetectionPipeline.cpp:250-266 computeDetectionWidth): search for objects of
    // at least `width` pixels — the packed 8-bit entries (operator()(packed ...), streamOpen / streamSubmit) reduce every frame by
    // scale = float(getWindowSize().width) / float(width) on the device (cv::resize: INTER_AREA when reducing, INTER_LINEAR else;
    // OpenCV's CV_8U arithmetic restated, parity unpinned: include/acf_hip.h) and map the boxes back with cv::Rect2f(o) * (1.f / scale).
    // width < 0 (default): off.
    void setMinObjectWidth(int width) { m_minObjectWidth = width; m_dirty = true; }
    int getMinObjectWidth() const { return m_minObjectWidth; }
    float inputScale() const { return m_minObjectWidth >= 0 ? float(getWindowSize().width) / float(m_minObjectWidth) : 1.f; }
    static Rect unscale(const Rect& o, float scale); // Resizer::operator()(objects) for one box
    int acfModify(const Modify& params); // acfModify.cpp:83-152

    // Detection: planar f32 transposed image (RGB in [0,1], or LUV after setIsLuv(true)); ACF.cpp:246-265.
    // Returns 0, APPENDS to objects; scores appended (no NMS) or assigned (NMS) exactly like ACF.cpp:332-364.
    int operator()(const MatP& IpTranspose, RectVec& objects, RealVec* scores = nullptr);
    // Packed interleaved RGB f32 in [0,1], upright rows x cols x 3 unless setIsTranspose(true) (ACF.cpp:135-141).
    int operator()(const float* rgbInterleaved, int rows, int cols, RectVec& objects, RealVec* scores = nullptr);
    // Packed 8-bit image (CV_8UC3 RGB / CV_8UC4 / CV_8UC1 of the reference's cv::Mat entry, ACF.cpp:135-141), upright
    // rows x cols, `pix` = ACF_HIP_PIX_*, `rowStrideBytes` 0 = tight.  /255, transpose, plane split and the colour
    // conversion all happen on the device (acf_hip_run_u8).
    int operator()(const uint8_t* packed, int rows, int cols, int pix, int rowStrideBytes, RectVec& objects, RealVec* scores = nullptr);
    // Multi-scale search on a pyramid (ACF.cpp:268-367).
    int operator()(const Pyramid& P, RectVec& objects, RealVec* scores = nullptr);
    // Batch of frames (no reference precedent; frames are independent): per-frame outputs.
    int detectBatch(const float* framesTransposedPlanar, int nFrames, int rows, int cols, int channels,
        std::vector<RectVec>& objects, std::vector<RealVec>* scores = nullptr);

    // Streaming front end (the role of GPUDetectionPipeline::runFast's two-frame FIFO, GPUDetectionPipeline.cpp:357-437):
    // batches of packed 8-bit frames are copied to the device while the previous batch computes.
    //   streamOpen(rows, cols, pix, stride, maxBatch, depth); t = streamSubmit(frames, n); streamCollect(t, objects, &scores);
    // `frames` should be page-locked (pinnedAlloc) and stay untouched until its ticket is collected; tickets are
    // collected in submission order.  At most maxDetectionsPerFrame (streamOpen) boxes per frame come back, in the
    // reference's order, before NMS/prune are applied exactly as in operator().
    void streamOpen(int rows, int cols, int pix, int rowStrideBytes, int maxBatch, int depth = 2, int maxDetectionsPerFrame = 4096);
    int streamSubmit(const uint8_t* frames, int nFrames);
    void streamCollect(int ticket, std::vector<RectVec>& objects, std::vector<RealVec>* scores = nullptr);
    void streamClose();
    static void* pinnedAlloc(size_t bytes);
    static void pinnedFree(void* p);

    void computePyramid(const MatP& Ip, Pyramid& P); // ACF.cpp:147-159
    // static int Detector::chnsCompute(const MatP&, const Options::Pyramid::Chns&, Channels&, bool isInit, const MatLoggerType&)
    // (ACF.h:342-349, chnsCompute.cpp:146-338): the channels of one image at its own scale (acf_hip_chns_compute; like the
    // reference's it needs no detector — it runs on a process-wide utility context on `device`).  isInit or an empty image:
    // only chns.pChns is filled (the reference's "return the estimate", :154-198).  With a logger the planes chnsCompute logs
    // (L, U, V; M; Mnorm, O; H — tags as in :241-250, gradientMag.cpp:119-123, :285-300, :322-329) are reported: the stages
    // then run one by one through the single-operator entries, whose results are the same floats.
    static int chnsCompute(const MatP& I, const Options::Pyramid::Chns& pChns, Channels& chns, bool isInit = false, const MatLoggerType& pLogger = {},
        int device = 0);
    // static void Detector::computeChannels(const MatP& Ip, MatP& Ip2, const MatLoggerType&) (ACF.h:419-420, ACF.cpp:183-240):
    // chnsCompute with the toolbox defaults, fused into one MatP [nChns planes stacked along rows] (fuseChannels, ACF.h:653-672)
    static void computeChannels(const MatP& Ip, MatP& Ip2, const MatLoggerType& pLogger = {}, int device = 0);
    // Detector::setLogger (ACF.h:578-581): operator()(MatP) then reports every pyramid level, transposed and stretched to
    // 0..255 (cv::normalize NORM_MINMAX to 8 bits, ACF.cpp:252-262; OpenCV's rounding restated: values are whole numbers), tag "%06d"
    void setLogger(MatLoggerType logger) { m_logger = std::move(logger); }
    // Detector::setStreamLogger (ACF.h:583-586): stored; like the reference's, nothing on the hot path writes to it
    void setStreamLogger(StreamLoggerType logger) { m_streamLogger = std::move(logger); }
    // The reference's OWN arithmetic at the three sites where its SSE kernels use _mm_rsqrt_ps / _mm_rcp_ps (include/acf_hip.h,
    // option "arith"): with the tables of a CPU installed the pyramid and the detections are what the reference's compiled
    // kernels give on that CPU, bit for bit (default: exact 1/sqrt, 1/x).  setReferenceArithmetic(true) probes the CPU this
    // process runs on — "what acf::Detector would return HERE" — and throws if its instructions are not table functions;
    // the second form installs given tables (4096 + 2 x 4096 entries, acf_hip_set_x86_tables).
    void setReferenceArithmetic(bool on);
    void setReferenceArithmetic(const uint32_t* rcp4096, const uint32_t* rsqrt8192);
    static bool probeHostArithmetic(std::vector<uint32_t>& rcp4096, std::vector<uint32_t>& rsqrt8192);
    // ... and for the static chnsCompute / computeChannels (their utility context on `device`): tables, or nullptr = exact again
    static void setChnsComputeReferenceArithmetic(const uint32_t* rcp4096, const uint32_t* rsqrt8192, int device = 0);
    // chnsPyramid.cpp:160-456.  With a logger, every real scale reports the planes chnsCompute hands to its logger, in its order
    // and with its tags (chnsCompute.cpp:241-250 L,U,V; gradientMag.cpp:119-123 M; chnsCompute.cpp:285-300 Mnorm, O; :322-329 H).
    int chnsPyramid(const MatP& I, const Options::Pyramid* pPyramid, Pyramid& pyramid, bool isInit = false, const MatLoggerType& logger = {});
    // chnsPyramid.cpp:461-529; sz = {width = image height, height = image width}
    static void getScales(int nPerOct, int nOctUp, const Size& minDs, int shrink, const Size& sz,
        std::vector<double>& scales, std::vector<Size2d>& scaleshw);

    // acfDetect1.cpp:309-335: one level of fused channels ([nChns * wP rows][hP cols]) ...
    void acfDetect1(const MatP& chns, int shrink, const Size& modelDsPad, int stride, double cascThr, DetectionVec& objects);
    // ... or of an atlas plane with one roi per channel (createDetector's computeChannelIndex branch, acfDetect1.cpp:262-265,
    // 346-366): the channels are gathered into a fused buffer on the host and take the same device path
    void acfDetect1(const MatP& atlas, const RectVec& rois, int shrink, const Size& modelDsPad, int stride, double cascThr, DetectionVec& objects);
    // the same on uint8_t channels ([nChns * wP rows][hP cols] bytes), the CV_8UC1 branch of allocDetector (acfDetect1.cpp:187-192)
    void acfDetect1(const uint8_t* chnsU8, int rows, int cols, DetectionVec& objects);
    // Detector::evaluate (ACF.h:543-544, acfDetect1.cpp:337-342): score of the window at (0,0) of a fused channel buffer, trees added
    // until the score drops to 0 or below (the reference fixes cascThr = 0 for this probe)
    float evaluate(const MatP& chns, int shrink, const Size& modelDsPad, int stride);
    // bbNms.cpp:229-304 (max / maxg / none), ObjectDetector.cpp:28-44
    static int bbNms(const DetectionVec& bbsIn, const Options::Nms& pNms, DetectionVec& bbs);
    void prune(RectVec& objects, RealVec& scores) const;

    // Single operators (static members of acf::Detector, ACF.h:441-493), on host planes.
    int rgbConvert(const MatP& I, MatP& J, const std::string& colorSpace);
    int convTri(const MatP& I, MatP& J, double r, bool inPlaceSemantics = false);
    int gradientMag(const MatP& I, MatP& M, MatP& O, int normRad, double normConst, int full);
    int gradientHist(const MatP& M, const MatP& O, MatP& H, int binSize, int nOrients, int full, int softBin = 0); // softBin even (gradientMex.cpp:391-509)

    acf_hip_ctx* context() { return m_ctx; }

private:
    void check(int rc, const char* what) const;
    void fillParams(acf_hip_params& p) const;
    void ensurePlan(int imgH, int imgW, int d, int batch);
    void fetch(int frame, RectVec& objects, RealVec* scores);
    void detect1(const float* f32, const uint8_t* u8, int rows, int cols, DetectionVec& objects);
    void finish(DetectionVec& bbs, RectVec& objects, RealVec* scores, bool forceHostNms = false) const; // ACF.cpp:332-364
    void logPyramid(); // ACF.cpp:252-262

    const hip::Api* m_api = nullptr;
    acf_hip_ctx* m_ctx = nullptr;
    bool m_good = false, m_dirty = true;
    bool m_doNms = false, m_isLuv = false, m_isTranspose = false, m_isRowMajor = false;
    void syncNms();
    bool m_nmsOnDevice = false;
    acf_hip_nms_params m_nmsSent{};
    size_t m_maxDetectionCount = 10;
    double m_detectionScorePruneRatio = 0.0;
    int m_planH = 0, m_planW = 0, m_planD = 0, m_planBatch = 0;
    std::vector<acf_hip_level> m_levels;
    int m_nChns = 0;
    std::vector<float> m_upright; // scratch for operator()(interleaved)
    int m_streamCap = 0, m_streamPix = -1, m_streamStride = 0;
    int m_minObjectWidth = -1, m_srcRows = 0, m_srcCols = 0; // frame size handed to the 8-bit entries (== the plan's without a resize)
    MatLoggerType m_logger;
    StreamLoggerType m_streamLogger;
    bool m_taps = false; // "taps" option on: per-stage planes stay readable (needed by the logger)
    void* m_pin = nullptr; // pinned scratch of operator()(packed 8-bit)
    size_t m_pinBytes = 0;
};

// One acf::HipDetector per visible MI355X (the north star's "batched frames shard across the GPUs of one node", host side in
// C++): detectBatch cuts the batch into contiguous blocks — the rule of acf_amd/dist.py:shard_range, blocks differ by at most
// one frame —, runs every block on its device from its own thread (frames are independent: no exchange between devices) and
// returns the per-frame results in frame order.  Setters are forwarded to every detector.
class HipDetectorPool
{
public:
    using RectVec = HipDetector::RectVec;
    using RealVec = HipDetector::RealVec;
    // devices: the device ordinals to use; empty = every visible gfx950 device (acf_hip_device_count)
    HipDetectorPool(const HipDetector::Options& o, const HipDetector::Classifier& c, std::vector<int> devices = {});
    explicit HipDetectorPool(const std::string& filename, std::vector<int> devices = {});
    size_t size() const { return m_dets.size(); }
    HipDetector& operator[](size_t i) { return *m_dets[i]; }
    static void shardRange(int nFrames, int world, int rank, int& begin, int& end);
    void setDoNonMaximaSuppression(bool f) { for (auto& d : m_dets) d->setDoNonMaximaSuppression(f); }
    void setMaxDetectionCount(size_t n) { for (auto& d : m_dets) d->setMaxDetectionCount(n); }
    void setDetectionScorePruneRatio(double r) { for (auto& d : m_dets) d->setDetectionScorePruneRatio(r); }
    void setIsLuv(bool f) { for (auto& d : m_dets) d->setIsLuv(f); }
    int acfModify(const HipDetector::Modify& m) { int rc = 0; for (auto& d : m_dets) rc |= d->acfModify(m); return rc; }
    int detectBatch(const float* framesTransposedPlanar, int nFrames, int rows, int cols, int channels,
        std::vector<RectVec>& objects, std::vector<RealVec>* scores = nullptr);

private:
    std::vector<std::unique_ptr<HipDetector>> m_dets;
};

} // namespace acf
