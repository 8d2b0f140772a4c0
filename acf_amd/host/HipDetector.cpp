#include "HipDetector.h"
#include "ModelIO.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <thread>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

namespace acf
{

static int colorSpaceFlag(const std::string& s)
{
    // rgbConvert.cpp:106-128
    if (s == "gray") return ACF_HIP_CS_GRAY;
    if (s == "rgb") return ACF_HIP_CS_RGB;
    if (s == "luv") return ACF_HIP_CS_LUV;
    if (s == "hsv") return ACF_HIP_CS_HSV;
    if (s == "orig") return ACF_HIP_CS_ORIG;
    throw Exception(ACF_HIP_E_INVALID, "unknown colorSpace: " + s);
}

HipDetector::HipDetector(const Options& o, const Classifier& c, int device)
{
    setModel(o, c, device);
}

// Detector(const std::string& filename) / Detector(std::istream&, hint) (ACF.cpp:38-46): good() reports the load status
HipDetector::HipDetector(const std::string& filename, int device)
{
    Options o;
    Classifier c;
    try
    {
        if (loadModelAny(filename, o, c) == 0)
        {
            setModel(o, c, device);
        }
    }
    catch (const Exception&)
    {
        m_good = false; // malformed file: like a failed deserialize
    }
}

HipDetector::HipDetector(std::istream& is, const std::string& hint, int device)
{
    Options o;
    Classifier c;
    try
    {
        if (hint.empty() || hint.find(".cpb") != std::string::npos) // ACFIO.cpp:218-231
        {
            loadCpb(is, o, c);
            setModel(o, c, device);
        }
        else if (loadAcfm(is, o, c))
        {
            setModel(o, c, device);
        }
    }
    catch (const Exception&)
    {
        m_good = false;
    }
}

HipDetector::~HipDetector()
{
    if (m_ctx && m_api)
    {
        m_api->acf_hip_destroy(m_ctx);
    }
    if (m_pin && m_api)
    {
        m_api->acf_hip_host_free(m_pin);
    }
}

void HipDetector::setDoParallel(bool flag)
{
    if (m_ctx && m_api)
    {
        check(m_api->acf_hip_set_option(m_ctx, "scale_streams", flag ? 1 : 0), "acf_hip_set_option");
    }
}

void HipDetector::setOption(const std::string& key, int value)
{
    if (m_ctx && m_api)
    {
        check(m_api->acf_hip_set_option(m_ctx, key.c_str(), value), "acf_hip_set_option");
    }
}

void HipDetector::check(int rc, const char* what) const
{
    if (rc != ACF_HIP_OK)
    {
        std::string msg = std::string(what) + ": ";
        msg += (m_ctx && m_api) ? m_api->acf_hip_last_error(m_ctx) : "no context";
        throw Exception(rc, msg);
    }
}

void HipDetector::setModel(const Options& o, const Classifier& c, int device)
{
    m_good = false;
    opts = o;
    clf = c;
    const size_t n = size_t(c.nTrees) * c.nTreeNodes;
    if (c.nTrees <= 0 || c.nTreeNodes <= 0 || c.fids.size() != n || c.thrs.size() != n || c.hs.size() != n ||
        (c.treeDepth == 0 && c.child.size() != n))
    {
        return; // like a failed deserialize: !good() (ACF.cpp:38-46)
    }
    m_api = &hip::load();
    if (!m_ctx)
    {
        if (m_api->acf_hip_create(device, nullptr, &m_ctx) != ACF_HIP_OK)
        {
            m_ctx = nullptr;
            return;
        }
    }
    m_dirty = true;
    m_planH = m_planW = 0;
    m_good = true;
}

void HipDetector::fillParams(acf_hip_params& p) const
{
    p = acf_hip_params{};
    p.nTrees = clf.nTrees;
    p.nTreeNodes = clf.nTreeNodes;
    p.treeDepth = clf.treeDepth;
    p.fids = clf.fids.data();
    p.thrs = clf.thrs.data();
    p.hs = clf.hs.data();
    p.child = clf.child.empty() ? nullptr : clf.child.data();
    // cv::Size {width = image-height axis, height = image-width axis} -> upright h / w
    p.modelDs_h = opts.modelDs.width;
    p.modelDs_w = opts.modelDs.height;
    p.modelDsPad_h = opts.modelDsPad.width;
    p.modelDsPad_w = opts.modelDsPad.height;
    p.stride = opts.stride;
    p.cascThr = opts.cascThr;
    const auto& py = opts.pPyramid;
    p.nPerOct = py.nPerOct;
    p.nOctUp = py.nOctUp;
    p.nApprox = py.nApprox;
    p.nLambdas = int(std::min<size_t>(py.lambdas.size(), 3));
    for (int i = 0; i < p.nLambdas; i++)
    {
        p.lambdas[i] = py.lambdas[i];
    }
    p.pad_h = py.pad.width;
    p.pad_w = py.pad.height;
    p.minDs_h = py.minDs.width;
    p.minDs_w = py.minDs.height;
    p.smooth = py.smooth;
    const auto& ch = py.pChns;
    p.shrink = ch.shrink;
    p.colorEnabled = ch.pColor.enabled;
    p.colorSmooth = ch.pColor.smooth;
    p.colorSpace = colorSpaceFlag(ch.pColor.colorSpace);
    p.gradMagEnabled = ch.pGradMag.enabled;
    p.colorChn = ch.pGradMag.colorChn;
    p.normRad = ch.pGradMag.normRad;
    p.normConst = ch.pGradMag.normConst;
    p.full = ch.pGradMag.full;
    p.gradHistEnabled = ch.pGradHist.enabled;
    p.binSize = ch.pGradHist.binSize;
    p.nOrients = ch.pGradHist.nOrients;
    p.softBin = ch.pGradHist.softBin;
    p.isLuv = (m_isLuv || ch.isLuv) ? 1 : 0;
    // LDCF post-stage (no reference counterpart; include/acf_hip.h): filters [k][nChns][5][5], MATLAB order of fs(:,:,c,f)
    p.ldcfK = opts.ldcfFilters.empty() ? 0 : opts.ldcfK;
    p.ldcfFilters = opts.ldcfFilters.empty() ? nullptr : opts.ldcfFilters.data();
}

int HipDetector::acfModify(const Modify& params)
{
    // acfModify.cpp:83-152.  The reference merges each override with
    // Field::merge, which only fills fields the model does NOT already have
    // (ACFField.h:54-60); a loaded model has them all, and this flattened
    // Options has them by construction, so — exactly as in the reference for a
    // complete model — the effects are the stride rounding (:139) and the
    // calibration offset on every hs entry (:143).
    const double shrink = opts.pPyramid.pChns.shrink;
    opts.stride = int(std::max(1.0, std::round(double(opts.stride) / shrink)) * shrink);
    if (params.has_cascCal)
    {
        for (auto& h : clf.hs)
        {
            h += float(params.cascCal);
        }
        opts.cascCal = params.cascCal;
    }
    m_dirty = true;
    return 0;
}

void HipDetector::ensurePlan(int imgH, int imgW, int d, int batch)
{
    if (!m_good)
    {
        throw Exception(ACF_HIP_E_NOMODEL, "HipDetector: no model");
    }
    if (m_dirty)
    {
        acf_hip_params p;
        fillParams(p);
        check(m_api->acf_hip_set_model(m_ctx, &p), "acf_hip_set_model");
        m_dirty = false;
        m_planH = 0;
    }
    if (imgH != m_planH || imgW != m_planW || d != m_planD || batch > m_planBatch)
    {
        check(m_api->acf_hip_plan(m_ctx, imgH, imgW, d, batch, 1 << 16), "acf_hip_plan");
        m_planH = imgH;
        m_planW = imgW;
        m_planD = d;
        m_planBatch = batch;
        m_streamCap = 0; // a re-plan closes the stream (its slots are sized by the plan)
        int n = 0;
        check(m_api->acf_hip_num_levels(m_ctx, &n, &m_nChns), "acf_hip_num_levels");
        m_levels.resize(size_t(n));
        check(m_api->acf_hip_get_levels(m_ctx, m_levels.data(), n), "acf_hip_get_levels");
    }
}

// ---- host-side bbNms + prune.  The hot paths run both on the device (acf_hip_set_nms: k_nms, one workgroup per frame,
// before the records leave the GPU); these host forms serve operator()(Pyramid) and callers of the public methods.
// Semantics: bbNms.cpp:111-192,229-304 and ObjectDetector.cpp:28-44, written as predicates over the score ranking:
//   a box is suppressed when some box RANKED ABOVE it overlaps it by more than `overlap` — any such box for "max",
//   one that itself survived for "maxg" (greedy).

namespace
{
struct RankedBox
{
    int x0, y0, x1, y1, area;
    size_t src; // index in the caller's list
};

// area of the intersection over the union / the smaller area (ovrDnm), 0 for disjoint boxes
inline double overlapRatio(const RankedBox& a, const RankedBox& b, bool overUnion)
{
    const int iw = std::min(a.x1, b.x1) - std::max(a.x0, b.x0);
    const int ih = std::min(a.y1, b.y1) - std::max(a.y0, b.y0);
    if (iw <= 0 || ih <= 0)
    {
        return 0.0;
    }
    const double inter = double(iw * ih);
    return inter / (overUnion ? double(a.area + b.area) - inter : double(std::min(a.area, b.area)));
}
} // namespace

void HipDetector::prune(RectVec& objects, RealVec& scores) const
{
    // keep the leading run of scores >= scores[0] * ratio, plus the first one below it, at most m_maxDetectionCount
    // (ObjectDetector.cpp:28-44: `cutoff = i + 1` is assigned before the ratio test)
    const size_t n = objects.size();
    if (n < 2)
    {
        return;
    }
    const size_t limit = std::min(m_maxDetectionCount, n);
    const double floorScore = scores[0] * m_detectionScorePruneRatio;
    size_t firstLow = 1;
    while (firstLow < limit && !(scores[firstLow] < floorScore))
    {
        firstLow++;
    }
    const size_t keep = limit < 2 ? 1 : std::min(firstLow + 1, limit);
    objects.resize(keep);
    scores.resize(keep);
}

int HipDetector::bbNms(const DetectionVec& bbsIn, const Options::Nms& pNms, DetectionVec& bbs)
{
    bbs = bbsIn;
    // "ms" and "cover" are pass-through stubs in the reference (bbNms.cpp:100-108)
    if (bbs.empty() || pNms.type == "none" || pNms.type == "ms" || pNms.type == "cover")
    {
        return 0;
    }
    if (pNms.type != "max" && pNms.type != "maxg")
    {
        throw Exception(ACF_HIP_E_INVALID, "bbNms: unknown type " + pNms.type);
    }
    if (pNms.ovrDnm != "union" && pNms.ovrDnm != "min")
    {
        throw Exception(ACF_HIP_E_INVALID, "bbNms: unknown ovrDnm " + pNms.ovrDnm);
    }
    const bool overUnion = pNms.ovrDnm == "union", greedy = pNms.type == "maxg";
    // ranking: scores at or above thr, best first; equal scores keep their input order (the reference's std::sort,
    // util/ordered.h:27, leaves that order open — this is one of its outcomes, and the device kernel's)
    std::vector<RankedBox> rank;
    rank.reserve(bbsIn.size());
    for (size_t i = 0; i < bbsIn.size(); i++)
    {
        if (!(bbsIn[i].score < pNms.thr))
        {
            const Rect& r = bbsIn[i].roi;
            rank.push_back(RankedBox{ r.x, r.y, r.x + r.width, r.y + r.height, r.width * r.height, i });
        }
    }
    std::stable_sort(rank.begin(), rank.end(), [&](const RankedBox& a, const RankedBox& b) { return bbsIn[a.src].score > bbsIn[b.src].score; });
    std::vector<char> survives(rank.size(), 1);
    bbs.clear();
    for (size_t j = 0; j < rank.size(); j++)
    {
        for (size_t i = 0; i < j && survives[j]; i++)
        {
            if ((!greedy || survives[i]) && overlapRatio(rank[i], rank[j], overUnion) > pNms.overlap)
            {
                survives[j] = 0;
            }
        }
        if (survives[j])
        {
            bbs.push_back(bbsIn[rank[j].src]);
        }
    }
    return 0;
}

// Tell the library what the reference's operator() does after the cascade (ACF.cpp:332-364): bbNms + prune on the
// device when non-maxima suppression is on, the raw list otherwise.
void HipDetector::syncNms()
{
    const bool deviceNms = m_doNms && (opts.pNms.type == "max" || opts.pNms.type == "maxg");
    if (!deviceNms)
    {
        if (m_nmsOnDevice)
        {
            check(m_api->acf_hip_set_nms(m_ctx, nullptr), "acf_hip_set_nms");
            m_nmsOnDevice = false;
        }
        return;
    }
    if (opts.pNms.ovrDnm != "union" && opts.pNms.ovrDnm != "min")
    {
        throw Exception(ACF_HIP_E_INVALID, "bbNms: unknown ovrDnm " + opts.pNms.ovrDnm);
    }
    acf_hip_nms_params q{};
    q.type = opts.pNms.type == "maxg" ? 2 : 1;
    q.ovrDnmUnion = opts.pNms.ovrDnm == "union";
    q.overlap = opts.pNms.overlap;
    q.thr = opts.pNms.thr;
    q.prune = 1;
    q.maxCount = int(std::min<size_t>(m_maxDetectionCount, size_t(1) << 30));
    q.pruneRatio = m_detectionScorePruneRatio;
    if (!m_nmsOnDevice || std::memcmp(&q, &m_nmsSent, sizeof(q)) != 0)
    {
        check(m_api->acf_hip_set_nms(m_ctx, &q), "acf_hip_set_nms");
        m_nmsSent = q;
        m_nmsOnDevice = true;
    }
}

void HipDetector::fetch(int frame, RectVec& objects, RealVec* scores)
{
    int n = 0;
    int rc = m_api->acf_hip_get_detections(m_ctx, frame, nullptr, 0, &n);
    // The device NMS takes ACF_HIP_NMS_CAP detections per frame; bbNms.cpp has no limit.  A frame beyond it (crowded scene,
    // low cascThr) is suppressed here from the raw list instead of failing.
    const bool hostNms = m_nmsOnDevice && rc == ACF_HIP_E_CAPACITY;
    auto get = hostNms ? m_api->acf_hip_get_raw_detections : m_api->acf_hip_get_detections;
    if (hostNms)
    {
        rc = get(m_ctx, frame, nullptr, 0, &n);
    }
    check(rc, "acf_hip_get_detections");
    std::vector<acf_hip_detection> d(size_t(std::max(n, 1)));
    check(get(m_ctx, frame, d.data(), n, &n), "acf_hip_get_detections");
    DetectionVec bbs(static_cast<size_t>(n));
    for (int i = 0; i < n; i++)
    {
        bbs[size_t(i)].roi = Rect(d[size_t(i)].x, d[size_t(i)].y, d[size_t(i)].w, d[size_t(i)].h);
        bbs[size_t(i)].score = double(d[size_t(i)].score);
    }
    finish(bbs, objects, scores, hostNms);
}

void HipDetector::finish(DetectionVec& bbs, RectVec& objects, RealVec* scores, bool forceHostNms) const
{
    if (m_doNms && (!m_nmsOnDevice || forceHostNms))
    {
        // ACF.cpp:332-353
        if (!bbs.empty())
        {
            DetectionVec out;
            bbNms(bbs, opts.pNms, out);
            RealVec bbScores;
            for (auto& b : out)
            {
                objects.push_back(b.roi);
                bbScores.push_back(b.score);
            }
            prune(objects, bbScores);
            if (scores)
            {
                *scores = bbScores;
            }
        }
    }
    else
    {
        // ACF.cpp:354-364
        for (auto& b : bbs)
        {
            objects.push_back(b.roi);
            if (scores)
            {
                scores->push_back(b.score);
            }
        }
    }
}

int HipDetector::operator()(const MatP& Ip, RectVec& objects, RealVec* scores)
{
    if (Ip.empty())
    {
        throw Exception(ACF_HIP_E_INVALID, "operator(): empty image");
    }
    // rows = image width, cols = image height
    ensurePlan(Ip.cols(), Ip.rows(), Ip.channels(), 1);
    syncNms();
    check(m_api->acf_hip_run_host(m_ctx, Ip.data(), 1), "acf_hip_run_host");
    if (m_logger)
    {
        logPyramid();
    }
    fetch(0, objects, scores);
    return 0;
}

// ACF.cpp:252-262: every level of the pyramid, transposed (upright: rows = nChns * hP ... of the fused buffer's transpose) and
// stretched to 0..255 as cv::normalize(d, canvas, 0, 255, NORM_MINMAX, CV_8UC1) does: (v - min) * (255 / (max - min)), rounded
// half to even and saturated (cv::saturate_cast<uchar>(cvRound)).  OpenCV's arithmetic restated: parity unpinned.
void HipDetector::logPyramid()
{
    for (size_t i = 0; i < m_levels.size(); i++)
    {
        const acf_hip_level& l = m_levels[i];
        const int rows = l.wP * m_nChns, cols = l.hP; // the fused buffer [nChns * wP rows][hP cols]
        MatP lev(rows, cols, 1);
        check(m_api->acf_hip_read_level(m_ctx, 0, int(i), lev.data()), "acf_hip_read_level");
        float lo = lev.data()[0], hi = lo;
        for (size_t k = 0; k < lev.numel(); k++)
        {
            lo = std::min(lo, lev.data()[k]);
            hi = std::max(hi, lev.data()[k]);
        }
        const double sc = hi > lo ? 255.0 / (double(hi) - double(lo)) : 0.0;
        MatP canvas(cols, rows, 1); // .t()
        for (int r = 0; r < rows; r++)
        {
            for (int c = 0; c < cols; c++)
            {
                const double v = (double(lev.data()[size_t(r) * cols + c]) - double(lo)) * sc;
                canvas.data()[size_t(c) * rows + r] = float(std::min(255.0, std::max(0.0, std::nearbyint(v))));
            }
        }
        char tag[16];
        std::snprintf(tag, sizeof(tag), "%06d", int(i));
        m_logger(canvas, tag);
    }
}

int HipDetector::operator()(const float* rgb, int rows, int cols, RectVec& objects, RealVec* scores)
{
    // ACF.cpp:135-141: transpose unless the caller already did, split interleaved -> planar (MatP.cpp:18-34)
    const int W = m_isTranspose ? rows : cols, H = m_isTranspose ? cols : rows;
    MatP Ip(W, H, 3);
    for (int z = 0; z < 3; z++)
    {
        float* P = Ip[z];
        for (int r = 0; r < rows; r++)
        {
            for (int c = 0; c < cols; c++)
            {
                const float v = rgb[(size_t(r) * cols + c) * 3 + z];
                if (m_isTranspose)
                {
                    P[size_t(r) * cols + c] = v; // already [W][H]
                }
                else
                {
                    P[size_t(c) * rows + r] = v;
                }
            }
        }
    }
    return (*this)(Ip, objects, scores);
}

int HipDetector::operator()(const uint8_t* packed, int rows, int cols, int pix, int rowStrideBytes, RectVec& objects, RealVec* scores)
{
    if (!packed || rows <= 0 || cols <= 0)
    {
        throw Exception(ACF_HIP_E_INVALID, "operator()(packed): empty image");
    }
    if (m_isTranspose)
    {
        throw Exception(ACF_HIP_E_UNSUPPORTED, "operator()(packed): 8-bit input must be upright (setIsTranspose(false))");
    }
    const int cpp = pix == ACF_HIP_PIX_GRAY ? 1 : (pix == ACF_HIP_PIX_RGBA || pix == ACF_HIP_PIX_BGRA) ? 4 : 3;
    const int stride = rowStrideBytes > 0 ? rowStrideBytes : cols * cpp;
    // one-frame stream: H2D on the copy stream, ingest + pyramid + cascade on the context's stream
    if (m_streamCap == 0 || m_srcRows != rows || m_srcCols != cols || m_planD != (cpp == 1 ? 1 : 3) || m_dirty || pix != m_streamPix ||
        stride != m_streamStride)
    {
        streamOpen(rows, cols, pix, stride, std::max(1, m_planBatch), 2);
    }
    const size_t bytes = size_t(stride) * rows;
    if (bytes > m_pinBytes)
    {
        pinnedFree(m_pin);
        m_pin = pinnedAlloc(bytes);
        m_pinBytes = bytes;
    }
    std::memcpy(m_pin, packed, bytes);
    const int t = streamSubmit(static_cast<const uint8_t*>(m_pin), 1);
    const int32_t* rec = nullptr;
    int n = 0;
    check(m_api->acf_hip_stream_collect(m_ctx, t, &rec, &n), "acf_hip_stream_collect");
    const size_t first = objects.size();
    fetch(0, objects, scores); // every detection, not only the first `cap` of the record
    if (m_minObjectWidth >= 0)
    {
        const float scale = inputScale();
        if (scale != 1.f) // Resizer::operator()(objects), acf.cpp:134-143
        {
            for (size_t i = first; i < objects.size(); i++)
            {
                objects[i] = unscale(objects[i], scale);
            }
        }
    }
    return 0;
}

Rect HipDetector::unscale(const Rect& o, float scale)
{
    // cv::Rect2f(o) * (1.f / scale) (acf.cpp:548-551: four float products) -> cv::Rect: saturate_cast<int> = cvRound of each field
    const float inv = 1.f / scale;
    Rect r;
    r.x = int(std::lrint(double(float(o.x) * inv)));
    r.y = int(std::lrint(double(float(o.y) * inv)));
    r.width = int(std::lrint(double(float(o.width) * inv)));
    r.height = int(std::lrint(double(float(o.height) * inv)));
    return r;
}

void HipDetector::streamOpen(int rows, int cols, int pix, int rowStrideBytes, int maxBatch, int depth, int maxDetectionsPerFrame)
{
    const int d = pix == ACF_HIP_PIX_GRAY ? 1 : 3;
    int planRows = rows, planCols = cols;
    const bool resize = m_minObjectWidth >= 0 && inputScale() != 1.f;
    if (resize)
    {
        check(m_api->acf_hip_resize_dims(rows, cols, double(inputScale()), &planRows, &planCols), "acf_hip_resize_dims");
    }
    ensurePlan(planRows, planCols, d, maxBatch);
    (void)m_api->acf_hip_stream_close(m_ctx); // (the input resize is set between the plan and the stream's buffers)
    check(m_api->acf_hip_set_input_resize(m_ctx, resize ? rows : 0, resize ? cols : 0, double(inputScale())), "acf_hip_set_input_resize");
    m_srcRows = rows;
    m_srcCols = cols;
    check(m_api->acf_hip_stream_open(m_ctx, pix, rowStrideBytes, maxDetectionsPerFrame, depth), "acf_hip_stream_open");
    m_streamCap = maxDetectionsPerFrame;
    m_streamPix = pix;
    m_streamStride = rowStrideBytes > 0 ? rowStrideBytes : cols * (pix == ACF_HIP_PIX_GRAY ? 1 : (pix == ACF_HIP_PIX_RGBA || pix == ACF_HIP_PIX_BGRA) ? 4 : 3);
}

int HipDetector::streamSubmit(const uint8_t* frames, int nFrames)
{
    int t = -1;
    syncNms();
    check(m_api->acf_hip_stream_submit(m_ctx, frames, nFrames, &t), "acf_hip_stream_submit");
    return t;
}

void HipDetector::streamCollect(int ticket, std::vector<RectVec>& objects, std::vector<RealVec>* scores)
{
    const int32_t* rec = nullptr;
    int n = 0;
    check(m_api->acf_hip_stream_collect(m_ctx, ticket, &rec, &n), "acf_hip_stream_collect");
    objects.assign(size_t(n), RectVec());
    if (scores)
    {
        scores->assign(size_t(n), RealVec());
    }
    const size_t per = 1 + 6 * size_t(m_streamCap);
    for (int f = 0; f < n; f++)
    {
        const int32_t* r = rec + size_t(f) * per;
        if (r[0] < 0)
        {
            // (the raw list of a streamed batch is gone once the slot is reused: no host fallback here)
            throw Exception(ACF_HIP_E_CAPACITY, "streamCollect: frame " + std::to_string(f) + " produced more than ACF_HIP_NMS_CAP raw detections, "
                "the capacity of the device NMS; detectBatch / operator() suppress such frames on the host, or raise cascThr");
        }
        if (r[0] > m_streamCap)
        {
            throw Exception(ACF_HIP_E_CAPACITY, "streamCollect: frame " + std::to_string(f) + " has " + std::to_string(r[0]) +
                " detections, more than maxDetectionsPerFrame = " + std::to_string(m_streamCap));
        }
        DetectionVec bbs(static_cast<size_t>(r[0]));
        for (int i = 0; i < r[0]; i++)
        {
            const int32_t* q = r + 1 + 6 * size_t(i);
            bbs[size_t(i)].roi = Rect(q[0], q[1], q[2], q[3]);
            float sc;
            std::memcpy(&sc, &q[4], sizeof(float));
            bbs[size_t(i)].score = double(sc);
        }
        finish(bbs, objects[size_t(f)], scores ? &(*scores)[size_t(f)] : nullptr);
        if (m_minObjectWidth >= 0 && inputScale() != 1.f)
        {
            for (Rect& o : objects[size_t(f)])
            {
                o = unscale(o, inputScale());
            }
        }
    }
}

void HipDetector::streamClose()
{
    if (m_ctx)
    {
        check(m_api->acf_hip_stream_close(m_ctx), "acf_hip_stream_close");
    }
    m_streamCap = 0;
}

void* HipDetector::pinnedAlloc(size_t bytes)
{
    void* p = nullptr;
    if (hip::load().acf_hip_host_alloc(bytes, &p))
    {
        throw Exception(ACF_HIP_E_HIP, "acf_hip_host_alloc");
    }
    return p;
}

void HipDetector::pinnedFree(void* p)
{
    if (p)
    {
        hip::load().acf_hip_host_free(p);
    }
}

int HipDetector::detectBatch(const float* frames, int nFrames, int rows, int cols, int channels,
    std::vector<RectVec>& objects, std::vector<RealVec>* scores)
{
    ensurePlan(cols, rows, channels, nFrames);
    syncNms();
    check(m_api->acf_hip_run_host(m_ctx, frames, nFrames), "acf_hip_run_host");
    objects.assign(size_t(nFrames), RectVec());
    if (scores)
    {
        scores->assign(size_t(nFrames), RealVec());
    }
    for (int f = 0; f < nFrames; f++)
    {
        fetch(f, objects[size_t(f)], scores ? &(*scores)[size_t(f)] : nullptr);
    }
    return 0;
}

void HipDetector::computePyramid(const MatP& Ip, Pyramid& P)
{
    chnsPyramid(Ip, &opts.pPyramid, P, true);
}

int HipDetector::chnsPyramid(const MatP& I, const Options::Pyramid* pPyramid, Pyramid& P, bool, const MatLoggerType& logger)
{
    if (pPyramid && pPyramid != &opts.pPyramid)
    {
        opts.pPyramid = *pPyramid;
        m_dirty = true;
    }
    if (logger && !m_taps)
    {
        // keep the per-stage planes (one more full-resolution write per real scale); takes effect at the next plan
        check(m_api->acf_hip_set_option(m_ctx, "taps", 1), "acf_hip_set_option(taps)");
        m_taps = true;
        m_planH = 0;
    }
    ensurePlan(I.cols(), I.rows(), I.channels(), 1);
    // upload + pyramid only
    {
        // run_host = H2D + pyramid + detect; the pyramid alone needs a device frame, so stage through run_host's
        // buffer by running the whole path (the cascade result is simply not fetched).
        check(m_api->acf_hip_run_host(m_ctx, I.data(), 1), "acf_hip_run_host");
    }
    P.clear();
    P.nScales = int(m_levels.size());
    P.nChns = m_nChns;
    P.nTypes = (opts.pPyramid.pChns.pColor.enabled ? 1 : 0) + (opts.pPyramid.pChns.pGradMag.enabled ? 1 : 0) +
        (opts.pPyramid.pChns.pGradHist.enabled ? 1 : 0);
    P.lambdas = opts.pPyramid.lambdas;
    if (P.lambdas.empty() && opts.pPyramid.nApprox > 0)
    {
        // estimated from this image (chnsPyramid.cpp:341-374)
        double lam[3] = { 0, 0, 0 };
        check(m_api->acf_hip_get_lambdas(m_ctx, 0, lam), "acf_hip_get_lambdas");
        P.lambdas.assign(lam, lam + 3);
    }
    P.data.resize(m_levels.size());
    for (size_t i = 0; i < m_levels.size(); i++)
    {
        const acf_hip_level& l = m_levels[i];
        P.scales.push_back(l.scale);
        Size2d s;
        s.width = l.scalehw_h; // cv::Size2d {width = image-height axis}
        s.height = l.scalehw_w;
        P.scaleshw.push_back(s);
        P.data[i].resize(1);
        P.data[i][0].create(l.wP * m_nChns, l.hP, 1); // fused: planes stacked along rows (ACF.h:653-672)
        check(m_api->acf_hip_read_level(m_ctx, 0, int(i), P.data[i][0].data()), "acf_hip_read_level");
    }
    if (logger)
    {
        const auto& ch = opts.pPyramid.pChns;
        const int d = ch.pColor.colorSpace == "gray" ? 1 : 3;
        const int nColor = ch.pColor.enabled ? d : 0;
        const int shrink = ch.shrink;
        int ordinal = 0;
        auto tag = [](const char* name, int cols, int rows) { return std::string(name) + ":" + std::to_string(cols) + "x" + std::to_string(rows); };
        for (size_t i = 0; i < m_levels.size(); i++)
        {
            const acf_hip_level& l = m_levels[i];
            if (!l.isReal)
            {
                continue;
            }
            const int h1 = l.hC * shrink, w1 = l.wC * shrink; // the real scale's image: planes are [w1 rows][h1 cols]
            MatP sm(w1, h1, d), one(w1, h1, 1);
            check(m_api->acf_hip_read_tap(m_ctx, 0, ACF_HIP_TAP_SMOOTHED, ordinal, sm.data(), int64_t(sm.numel())), "acf_hip_read_tap");
            static const char* luv[3] = { "L", "U", "V" };
            for (int z = 0; z < d; z++)
            {
                MatP plane(w1, h1, 1, sm[z]);
                logger(plane, tag(d == 3 ? luv[z] : "L", h1, w1));
            }
            if (ch.pGradMag.enabled || ch.pGradHist.enabled)
            {
                for (const auto& t : { std::make_pair(int(ACF_HIP_TAP_M), "M"), std::make_pair(int(ACF_HIP_TAP_MNORM), "Mnorm"), std::make_pair(int(ACF_HIP_TAP_O), "O") })
                {
                    check(m_api->acf_hip_read_tap(m_ctx, 0, t.first, ordinal, one.data(), int64_t(one.numel())), "acf_hip_read_tap");
                    logger(one, tag(t.second, h1, w1));
                }
            }
            if (ch.pGradHist.enabled)
            {
                // cv::hconcat of the nOrients histogram planes (chnsCompute.cpp:322-329): [wC rows][nOrients * hC cols]
                const int nO = ch.pGradHist.nOrients;
                MatP raw(l.wC * m_nChns, l.hC, 1);
                check(m_api->acf_hip_read_tap(m_ctx, 0, ACF_HIP_TAP_CHNS, int(i), raw.data(), int64_t(raw.numel())), "acf_hip_read_tap");
                const int first = nColor + (ch.pGradMag.enabled ? 1 : 0);
                MatP hc(l.wC, nO * l.hC, 1);
                for (int b = 0; b < nO; b++)
                {
                    for (int r = 0; r < l.wC; r++)
                    {
                        std::memcpy(hc.data() + (size_t(r) * nO + b) * l.hC, raw.data() + (size_t(first + b) * l.wC + r) * l.hC, sizeof(float) * l.hC);
                    }
                }
                logger(hc, tag("H", nO * l.hC, l.wC));
            }
            ordinal++;
        }
    }
    return 0;
}

int HipDetector::operator()(const Pyramid& P, RectVec& objects, RealVec* scores)
{
    if (P.nScales != int(m_levels.size()) || P.nScales == 0)
    {
        throw Exception(ACF_HIP_E_NOPLAN, "operator()(Pyramid): pyramid was not produced by this detector");
    }
    // Always the pyramid that was handed in — like the reference, which re-runs acfDetect1 on every level of P
    // (ACF.cpp:268-367): one acfDetect1 per level + box mapping.  (Round 1 short-cut to the device-resident result of the
    // last run when P carried a matching generation tag; another image, acfModify or an edit of P.data in between then
    // returned detections that did not belong to P.)
    const int shift_w = (opts.modelDsPad.width - opts.modelDs.width) / 2 - opts.pPyramid.pad.width;   // image-height axis
    const int shift_h = (opts.modelDsPad.height - opts.modelDs.height) / 2 - opts.pPyramid.pad.height; // image-width axis
    DetectionVec all;
    for (int i = 0; i < P.nScales; i++)
    {
        DetectionVec ds;
        if (P.rois.size() > size_t(i) && !P.rois[size_t(i)].empty()) // ACF.cpp:292-299
        {
            acfDetect1(P.data[size_t(i)][0], P.rois[size_t(i)], opts.pPyramid.pChns.shrink, opts.modelDsPad, opts.stride, opts.cascThr, ds);
        }
        else
        {
            acfDetect1(P.data[size_t(i)][0], opts.pPyramid.pChns.shrink, opts.modelDsPad, opts.stride, opts.cascThr, ds);
        }
        // ACF.cpp:302-312 — ds rois are in the transposed convention (x along image-y)
        const double sw = P.scaleshw[size_t(i)].width, sh = P.scaleshw[size_t(i)].height;
        const int bw = int(std::nearbyint(double(opts.modelDs.width) / P.scales[size_t(i)]));
        const int bh = int(std::nearbyint(double(opts.modelDs.height) / P.scales[size_t(i)]));
        for (auto& d : ds)
        {
            const int x = int(double(d.roi.x + shift_w) / sw);
            const int y = int(double(d.roi.y + shift_h) / sh);
            Detection o;
            o.roi = Rect(y, x, bh, bw); // swap back to upright (ACF.cpp:310-311)
            o.score = d.score;
            all.push_back(o);
        }
    }
    if (m_doNms)
    {
        if (!all.empty())
        {
            DetectionVec out;
            bbNms(all, opts.pNms, out);
            RealVec bbScores;
            for (auto& b : out)
            {
                objects.push_back(b.roi);
                bbScores.push_back(b.score);
            }
            prune(objects, bbScores);
            if (scores)
            {
                *scores = bbScores;
            }
        }
    }
    else
    {
        for (auto& b : all)
        {
            objects.push_back(b.roi);
            if (scores)
            {
                scores->push_back(b.score);
            }
        }
    }
    return 0;
}

void HipDetector::acfDetect1(const MatP& chns, int, const Size&, int, double, DetectionVec& objects)
{
    // chns: fused level buffer, rows = nChns * wP, cols = hP
    detect1(chns.data(), nullptr, chns.rows(), chns.cols(), objects);
}

void HipDetector::acfDetect1(const MatP& atlas, const RectVec& rois, int, const Size&, int, double, DetectionVec& objects)
{
    // computeChannelIndex (acfDetect1.cpp:346-366): cids[z][c][r] = z * chnStride + c * rowStride + r with
    // chnStride = rois[1].x - rois[0].x and rowStride = the atlas plane's step; window grid from the channel size
    if (rois.size() < 2)
    {
        throw Exception(ACF_HIP_E_INVALID, "acfDetect1(rois): at least two channel rois (the reference asserts rois.size() > 1)");
    }
    const int nChns = int(rois.size()), rowStride = atlas.cols();
    const int chnStride = rois[1].x - rois[0].x;
    const int wP = rois[0].height, hP = rois[0].width; // a channel: wP columns c (stride rowStride) of hP rows r
    if (chnStride <= 0 || wP <= 0 || hP <= 0 || hP > rowStride ||
        size_t(nChns - 1) * chnStride + size_t(wP - 1) * rowStride + hP > atlas.numel())
    {
        throw Exception(ACF_HIP_E_INVALID, "acfDetect1(rois): channel rois outside the atlas plane");
    }
    MatP fused(nChns * wP, hP, 1);
    for (int z = 0; z < nChns; z++)
    {
        for (int c = 0; c < wP; c++)
        {
            std::memcpy(fused.data() + (size_t(z) * wP + c) * hP, atlas.data() + size_t(z) * chnStride + size_t(c) * rowStride, sizeof(float) * hP);
        }
    }
    detect1(fused.data(), nullptr, fused.rows(), fused.cols(), objects);
}

void HipDetector::acfDetect1(const uint8_t* chnsU8, int rows, int cols, DetectionVec& objects)
{
    // the CV_8UC1 case of allocDetector (acfDetect1.cpp:187-192) with Classifier::thrsU8 (getScaledThresholds :168-182)
    detect1(nullptr, chnsU8, rows, cols, objects);
}

void HipDetector::detect1(const float* f32, const uint8_t* u8, int rows, int cols, DetectionVec& objects)
{
    if (!m_good)
    {
        throw Exception(ACF_HIP_E_NOMODEL, "acfDetect1: no model");
    }
    if (m_dirty)
    {
        acf_hip_params p;
        fillParams(p);
        check(m_api->acf_hip_set_model(m_ctx, &p), "acf_hip_set_model");
        m_dirty = false;
        m_planH = 0;
    }
    const auto& ch = opts.pPyramid.pChns;
    const int nColor = ch.pColor.enabled ? (ch.pColor.colorSpace == "gray" ? 1 : 3) : 0;
    const int nC = nColor + (ch.pGradMag.enabled ? 1 : 0) + (ch.pGradHist.enabled ? ch.pGradHist.nOrients : 0);
    const int wP = rows / nC, hP = cols;
    const int cap = std::max(1, wP * hP);
    std::vector<acf_hip_hit> hits(static_cast<size_t>(cap));
    int n = 0;
    if (u8)
    {
        if (clf.thrsU8.empty())
        {
            // what the loader does once (ACFIOArchive.h:96-99)
            clf.thrsU8.resize(clf.thrs.size());
            check(m_api->acf_hip_thrs_u8(clf.thrs.data(), int(clf.thrs.size()), clf.thrsU8.data()), "acf_hip_thrs_u8");
        }
        check(m_api->acf_hip_op_acf_detect1_u8(m_ctx, u8, hP, wP, nC, clf.thrsU8.data(), hits.data(), cap, &n), "acf_hip_op_acf_detect1_u8");
    }
    else
    {
        check(m_api->acf_hip_op_acf_detect1(m_ctx, f32, hP, wP, nC, hits.data(), cap, &n), "acf_hip_op_acf_detect1");
    }
    for (int i = 0; i < n; i++)
    {
        Detection d;
        // acfDetect1.cpp:326-332: Rect(c*stride, r*stride, modelWd, modelHt) then swapped into the transposed convention
        d.roi = Rect(hits[size_t(i)].r * opts.stride, hits[size_t(i)].c * opts.stride, opts.modelDsPad.width, opts.modelDsPad.height);
        d.score = double(hits[size_t(i)].score);
        objects.push_back(d);
    }
}

float HipDetector::evaluate(const MatP& chns, int, const Size&, int)
{
    if (!m_good)
    {
        throw Exception(ACF_HIP_E_NOMODEL, "evaluate: no model");
    }
    if (m_dirty)
    {
        acf_hip_params p;
        fillParams(p);
        check(m_api->acf_hip_set_model(m_ctx, &p), "acf_hip_set_model");
        m_dirty = false;
        m_planH = 0;
    }
    const auto& ch = opts.pPyramid.pChns;
    const int nColor = ch.pColor.enabled ? (ch.pColor.colorSpace == "gray" ? 1 : 3) : 0;
    const int nC = nColor + (ch.pGradMag.enabled ? 1 : 0) + (ch.pGradHist.enabled ? ch.pGradHist.nOrients : 0);
    float score = 0.f;
    check(m_api->acf_hip_op_evaluate(m_ctx, chns.data(), chns.cols(), chns.rows() / nC, nC, 0.0, &score), "acf_hip_op_evaluate");
    return score;
}

void HipDetector::getScales(int nPerOct, int nOctUp, const Size& minDs, int shrink, const Size& sz,
    std::vector<double>& scales, std::vector<Size2d>& scaleshw)
{
    const hip::Api& api = hip::load();
    int n = 0;
    std::vector<double> s(512), a(512), b(512);
    const int rc = api.acf_hip_get_scales(nPerOct, nOctUp, minDs.width, minDs.height, shrink, sz.width, sz.height, s.data(), a.data(), b.data(), 512, &n);
    if (rc != ACF_HIP_OK)
    {
        throw Exception(rc, "getScales");
    }
    scales.assign(s.begin(), s.begin() + n);
    scaleshw.resize(size_t(n));
    for (int i = 0; i < n; i++)
    {
        scaleshw[size_t(i)].width = a[size_t(i)];
        scaleshw[size_t(i)].height = b[size_t(i)];
    }
}

int HipDetector::rgbConvert(const MatP& I, MatP& J, const std::string& colorSpace)
{
    const int flag = colorSpaceFlag(colorSpace);
    if (!m_ctx)
    {
        m_api = &hip::load();
        check(m_api->acf_hip_create(0, nullptr, &m_ctx), "acf_hip_create");
    }
    J.create(I.rows(), I.cols(), flag == ACF_HIP_CS_GRAY ? 1 : 3);
    check(m_api->acf_hip_op_rgb_convert(m_ctx, I.data(), J.data(), I.cols(), I.rows(), flag), "acf_hip_op_rgb_convert");
    return 0;
}

int HipDetector::convTri(const MatP& I, MatP& J, double r, bool inPlaceSemantics)
{
    if (!m_ctx)
    {
        m_api = &hip::load();
        check(m_api->acf_hip_create(0, nullptr, &m_ctx), "acf_hip_create");
    }
    J.create(I.rows(), I.cols(), I.channels());
    check(m_api->acf_hip_op_conv_tri(m_ctx, I.data(), J.data(), I.cols(), I.rows(), I.channels(), r, inPlaceSemantics ? 1 : 0), "acf_hip_op_conv_tri");
    return 0;
}

int HipDetector::gradientMag(const MatP& I, MatP& M, MatP& O, int normRad, double normConst, int full)
{
    if (!m_ctx)
    {
        m_api = &hip::load();
        check(m_api->acf_hip_create(0, nullptr, &m_ctx), "acf_hip_create");
    }
    M.create(I.rows(), I.cols(), 1);
    O.create(I.rows(), I.cols(), 1);
    check(m_api->acf_hip_op_gradient_mag(m_ctx, I.data(), M.data(), O.data(), nullptr, I.cols(), I.rows(), normRad, normConst, full), "acf_hip_op_gradient_mag");
    return 0;
}

int HipDetector::gradientHist(const MatP& M, const MatP& O, MatP& H, int binSize, int nOrients, int full, int softBin)
{
    if (!m_ctx)
    {
        m_api = &hip::load();
        check(m_api->acf_hip_create(0, nullptr, &m_ctx), "acf_hip_create");
    }
    H.create(M.rows() / binSize, M.cols() / binSize, nOrients);
    check(m_api->acf_hip_op_gradient_hist(m_ctx, M.data(), O.data(), H.data(), M.cols(), M.rows(), binSize, nOrients, softBin, full), "acf_hip_op_gradient_hist");
    return 0;
}

// ---- chnsCompute / computeChannels as the reference's static functions: a process-wide utility context per device
namespace
{
struct UtilCtx
{
    std::mutex m;
    std::vector<acf_hip_ctx*> ctx; // by device ordinal
};
UtilCtx& utilCtx()
{
    static UtilCtx u; // (contexts live until the process ends, like the reference's function-static tables)
    return u;
}
acf_hip_ctx* utilContext(const hip::Api& api, int device)
{
    UtilCtx& u = utilCtx();
    if (device < 0)
    {
        throw Exception(ACF_HIP_E_INVALID, "chnsCompute: device");
    }
    if (size_t(device) >= u.ctx.size())
    {
        u.ctx.resize(size_t(device) + 1, nullptr);
    }
    if (!u.ctx[size_t(device)])
    {
        const int rc = api.acf_hip_create(device, nullptr, &u.ctx[size_t(device)]);
        if (rc != ACF_HIP_OK)
        {
            u.ctx[size_t(device)] = nullptr;
            throw Exception(rc, "chnsCompute: acf_hip_create failed (no gfx950 device?)");
        }
    }
    return u.ctx[size_t(device)];
}
void utilCheck(const hip::Api& api, acf_hip_ctx* c, int rc, const char* what)
{
    if (rc != ACF_HIP_OK)
    {
        throw Exception(rc, std::string(what) + ": " + api.acf_hip_last_error(c));
    }
}
void chnsParams(const HipDetector::Options::Pyramid::Chns& ch, acf_hip_params& p)
{
    p = acf_hip_params{};
    p.shrink = ch.shrink;
    p.colorEnabled = ch.pColor.enabled;
    p.colorSmooth = ch.pColor.smooth;
    p.colorSpace = colorSpaceFlag(ch.pColor.colorSpace);
    p.gradMagEnabled = ch.pGradMag.enabled;
    p.colorChn = ch.pGradMag.colorChn;
    p.normRad = ch.pGradMag.normRad;
    p.normConst = ch.pGradMag.normConst;
    p.full = ch.pGradMag.full;
    p.gradHistEnabled = ch.pGradHist.enabled;
    p.binSize = ch.pGradHist.binSize;
    p.nOrients = ch.pGradHist.nOrients;
    p.softBin = ch.pGradHist.softBin;
    p.isLuv = ch.isLuv ? 1 : 0;
}
std::string planeTag(const char* name, int cols, int rows)
{
    return std::string(name) + ":" + std::to_string(cols) + "x" + std::to_string(rows);
}
} // namespace

int HipDetector::chnsCompute(const MatP& IIn, const Options::Pyramid::Chns& pChns, Channels& chns, bool isInit, const MatLoggerType& pLogger, int device)
{
    chns.pChns = pChns; // (the flattened Options tree is complete by construction: the merge of chnsCompute.cpp:154-195 has nothing to fill)
    if (isInit || IIn.empty())
    {
        return 0; // "return the estimate" (:197-198)
    }
    const hip::Api& api = hip::load();
    UtilCtx& u = utilCtx();
    std::lock_guard<std::mutex> lock(u.m);
    acf_hip_ctx* c = utilContext(api, device);
    acf_hip_params p;
    chnsParams(pChns, p);
    const int h = IIn.cols(), w = IIn.rows(), d = IIn.channels(); // rows = image width, cols = image height
    int nC = 0, hc = 0, wc = 0;
    utilCheck(api, c, api.acf_hip_chns_compute(c, &p, IIn.data(), h, w, d, nullptr, 0, &nC, &hc, &wc), "acf_hip_chns_compute");
    const int dcol = p.colorSpace == ACF_HIP_CS_GRAY ? 1 : 3;
    std::vector<float> all(size_t(nC) * hc * wc);
    if (!pLogger)
    {
        utilCheck(api, c, api.acf_hip_chns_compute(c, &p, IIn.data(), h, w, d, all.data(), int64_t(all.size()), &nC, &hc, &wc), "acf_hip_chns_compute");
    }
    else
    {
        // stage by stage through the single-operator entries, reporting what chnsCompute reports
        const int shrink = p.shrink, H = h - h % shrink, W = w - w % shrink;
        const size_t np = size_t(H) * W, ns = size_t(hc) * wc;
        const int dImg = d == 5 ? 3 : d; // (five planes: the image's own M, O behind three image planes, chnsCompute.cpp:219-226)
        MatP I(W, H, d);
        for (int z = 0; z < d; z++)
        {
            for (int x = 0; x < W; x++)
            {
                std::memcpy(I[z] + size_t(x) * H, IIn[z] + size_t(x) * h, sizeof(float) * H); // the crop (:203-217)
            }
        }
        MatP col(W, H, dcol);
        const bool pass = dImg == 3 && (p.colorSpace == ACF_HIP_CS_ORIG || p.colorSpace == ACF_HIP_CS_RGB || (p.isLuv && p.colorSpace == ACF_HIP_CS_LUV));
        if (pass)
        {
            std::memcpy(col.data(), I.data(), sizeof(float) * 3 * np);
        }
        else if (d == 1 && p.colorSpace == ACF_HIP_CS_ORIG)
        {
            for (int z = 0; z < 3; z++)
            {
                std::memcpy(col[z], I.data(), sizeof(float) * np); // (a 1-plane image replicated: chnsPyramid.cpp:242-243)
            }
        }
        else if (d == 1)
        {
            MatP rep(W, H, 3);
            for (int z = 0; z < 3; z++)
            {
                std::memcpy(rep[z], I.data(), sizeof(float) * np);
            }
            utilCheck(api, c, api.acf_hip_op_rgb_convert(c, rep.data(), col.data(), H, W, p.colorSpace), "acf_hip_op_rgb_convert");
        }
        else
        {
            utilCheck(api, c, api.acf_hip_op_rgb_convert(c, I.data(), col.data(), H, W, p.colorSpace), "acf_hip_op_rgb_convert");
        }
        if (p.colorSmooth > 0)
        {
            MatP sm(W, H, dcol);
            utilCheck(api, c, api.acf_hip_op_conv_tri(c, col.data(), sm.data(), H, W, dcol, p.colorSmooth, 1), "acf_hip_op_conv_tri");
            col = sm;
        }
        static const char* luv[3] = { "L", "U", "V" };
        for (int z = 0; z < dcol; z++)
        {
            MatP plane(W, H, 1, col[z]);
            pLogger(plane, planeTag(dcol == 3 ? luv[z] : "L", H, W));
        }
        float* o = all.data();
        if (p.colorEnabled)
        {
            utilCheck(api, c, api.acf_hip_op_im_resample(c, col.data(), o, H, W, hc, wc, dcol, 1.0), "acf_hip_op_im_resample");
            o += ns * dcol;
        }
        if (p.gradMagEnabled || p.gradHistEnabled)
        {
            MatP M(W, H, 1), O(W, H, 1), Mn(W, H, 1), On(W, H, 1);
            if (d == 5)
            {
                // M = MO[0], O = MO[1] (:265-269): nothing computed, nothing logged for them
                std::memcpy(Mn.data(), I[3], sizeof(float) * np);
                std::memcpy(On.data(), I[4], sizeof(float) * np);
            }
            else
            {
                utilCheck(api, c, api.acf_hip_op_gradient_mag(c, col[p.colorChn], M.data(), O.data(), nullptr, H, W, 0, p.normConst, p.full), "acf_hip_op_gradient_mag");
                pLogger(M, planeTag("M", H, W)); // gradientMag.cpp:119-123: before the normalisation
                utilCheck(api, c, api.acf_hip_op_gradient_mag(c, col[p.colorChn], Mn.data(), On.data(), nullptr, H, W, p.normRad, p.normConst, p.full), "acf_hip_op_gradient_mag");
                pLogger(Mn, planeTag("Mnorm", H, W));
                pLogger(On, planeTag("O", H, W));
            }
            if (p.gradMagEnabled)
            {
                utilCheck(api, c, api.acf_hip_op_im_resample(c, Mn.data(), o, H, W, hc, wc, 1, 1.0), "acf_hip_op_im_resample");
                o += ns;
            }
            if (p.gradHistEnabled)
            {
                utilCheck(api, c, api.acf_hip_op_gradient_hist(c, Mn.data(), On.data(), o, H, W, shrink, p.nOrients, p.softBin, p.full), "acf_hip_op_gradient_hist");
                MatP hcat(wc, p.nOrients * hc, 1); // cv::hconcat of the planes (:322-329)
                for (int b = 0; b < p.nOrients; b++)
                {
                    for (int r = 0; r < wc; r++)
                    {
                        std::memcpy(hcat.data() + (size_t(r) * p.nOrients + b) * hc, o + (size_t(b) * wc + r) * hc, sizeof(float) * hc);
                    }
                }
                pLogger(hcat, planeTag("H", p.nOrients * hc, wc));
            }
        }
    }
    // addChn (:340-370): one entry per enabled type
    chns.data.clear();
    chns.info.clear();
    const float* o = all.data();
    auto add = [&](int n, const char* name, const char* padWith) {
        MatP m(wc, hc, n);
        std::memcpy(m.data(), o, sizeof(float) * m.numel());
        o += m.numel();
        chns.data.push_back(m);
        Channels::Info info;
        info.name = name;
        info.nChns = n;
        info.padWith = padWith;
        chns.info.push_back(info);
    };
    if (p.colorEnabled)
    {
        add(dcol, "color channels", "replicate");
    }
    if (p.gradMagEnabled)
    {
        add(1, "gradient magnitude", "");
    }
    if (p.gradHistEnabled)
    {
        add(p.nOrients, "gradient histogram", "");
    }
    chns.nTypes = int(chns.data.size());
    return 0;
}

void HipDetector::computeChannels(const MatP& Ip, MatP& Ip2, const MatLoggerType& pLogger, int device)
{
    // ACF.cpp:185-232: the toolbox defaults (shrink 4; colour luv, smooth 1; gradMag normRad 5, normConst .005; 6 orientations)
    Options::Pyramid::Chns dfs;
    Channels chns;
    chnsCompute(Ip, dfs, chns, false, pLogger, device);
    // fuseChannels (ACF.h:653-672): every plane of every type stacked along rows
    int rows = 0, cols = 0, planes = 0;
    for (const MatP& m : chns.data)
    {
        rows = m.rows();
        cols = m.cols();
        planes += m.channels();
    }
    Ip2.create(rows * planes, cols, 1);
    float* o = Ip2.data();
    for (const MatP& m : chns.data)
    {
        std::memcpy(o, m.data(), sizeof(float) * m.numel());
        o += m.numel();
    }
}

// ---- the reference's arithmetic (option "arith")
namespace
{
// the table functions of include/acf_hip.h, for the check of a probed CPU
inline uint32_t tblRcp(uint32_t u, const uint32_t* T)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e == 0xffu)
    {
        return m ? (u | 0x400000u) : s;
    }
    if (e == 0)
    {
        return s | 0x7f800000u;
    }
    const uint32_t t = T[m >> 11];
    const int re = int((t >> 23) & 0xffu) + 127 - int(e);
    return re <= 0 ? s : (s | (uint32_t(re) << 23) | (t & 0x7fffffu));
}
inline uint32_t tblRsqrt(uint32_t u, const uint32_t* T)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e == 0xffu)
    {
        return m ? (u | 0x400000u) : (s ? 0xffc00000u : 0u);
    }
    if (e == 0)
    {
        return s | 0x7f800000u;
    }
    if (s)
    {
        return 0xffc00000u;
    }
    const int ue = int(e) - 127, odd = ue & 1, half = (ue - odd) / 2;
    const uint32_t t = T[(odd << 12) | int(m >> 11)];
    return (uint32_t(int((t >> 23) & 0xffu) - half) << 23) | (t & 0x7fffffu);
}
#if defined(__SSE__)
inline uint32_t hwRcp(uint32_t u)
{
    float f, o;
    std::memcpy(&f, &u, 4);
    o = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(f)));
    std::memcpy(&u, &o, 4);
    return u;
}
inline uint32_t hwRsqrt(uint32_t u)
{
    float f, o;
    std::memcpy(&f, &u, 4);
    o = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(f)));
    std::memcpy(&u, &o, 4);
    return u;
}
#endif
} // namespace

bool HipDetector::probeHostArithmetic(std::vector<uint32_t>& rcp, std::vector<uint32_t>& rsq)
{
#if defined(__SSE__)
    rcp.resize(4096);
    rsq.resize(8192);
    for (uint32_t i = 0; i < 4096; i++)
    {
        rcp[i] = hwRcp((127u << 23) | (i << 11));
        rsq[i] = hwRsqrt((127u << 23) | (i << 11));
        rsq[4096 + i] = hwRsqrt((128u << 23) | (i << 11));
    }
    // are this CPU's instructions those table functions?  2^22 inputs spread over all bit patterns, every mantissa of one binade,
    // and the exponent range's two ends (tests/golden/make_x86_tables.py runs the same comparison over all 2^32)
    auto same = [&](uint32_t u) { return hwRcp(u) == tblRcp(u, rcp.data()) && hwRsqrt(u) == tblRsqrt(u, rsq.data()); };
    for (uint64_t i = 0; i < (1ull << 22); i++)
    {
        if (!same(uint32_t(i * 1021u)))
        {
            return false;
        }
    }
    for (uint32_t m = 0; m < (1u << 23); m += 7)
    {
        if (!same((127u << 23) | m) || !same((128u << 23) | m))
        {
            return false;
        }
    }
    for (uint32_t k = 0; k < (1u << 16); k++)
    {
        if (!same(k * 257u) || !same(0x7e800000u + k * 513u) || !same(0x80000000u + k * 257u))
        {
            return false;
        }
    }
    return true;
#else
    (void)rcp;
    (void)rsq;
    return false;
#endif
}

void HipDetector::setReferenceArithmetic(const uint32_t* rcp4096, const uint32_t* rsqrt8192)
{
    if (!m_ctx)
    {
        m_api = &hip::load();
        check(m_api->acf_hip_create(0, nullptr, &m_ctx), "acf_hip_create");
    }
    check(m_api->acf_hip_set_x86_tables(m_ctx, rcp4096, rsqrt8192), "acf_hip_set_x86_tables");
    check(m_api->acf_hip_set_option(m_ctx, "arith", 1), "acf_hip_set_option(arith)");
}

void HipDetector::setChnsComputeReferenceArithmetic(const uint32_t* rcp4096, const uint32_t* rsqrt8192, int device)
{
    const hip::Api& api = hip::load();
    UtilCtx& u = utilCtx();
    std::lock_guard<std::mutex> lock(u.m);
    acf_hip_ctx* c = utilContext(api, device);
    if (rcp4096 && rsqrt8192)
    {
        utilCheck(api, c, api.acf_hip_set_x86_tables(c, rcp4096, rsqrt8192), "acf_hip_set_x86_tables");
        utilCheck(api, c, api.acf_hip_set_option(c, "arith", 1), "acf_hip_set_option(arith)");
    }
    else
    {
        utilCheck(api, c, api.acf_hip_set_option(c, "arith", 0), "acf_hip_set_option(arith)");
    }
}

void HipDetector::setReferenceArithmetic(bool on)
{
    if (!on)
    {
        if (m_ctx && m_api)
        {
            check(m_api->acf_hip_set_option(m_ctx, "arith", 0), "acf_hip_set_option(arith)");
        }
        return;
    }
    std::vector<uint32_t> rcp, rsq;
    if (!probeHostArithmetic(rcp, rsq))
    {
        throw Exception(ACF_HIP_E_UNSUPPORTED, "setReferenceArithmetic: this CPU's rcpps / rsqrtps are not functions of the top 12 mantissa bits (or not an SSE host)");
    }
    setReferenceArithmetic(rcp.data(), rsq.data());
}

// ---- HipDetectorPool: one detector per device, frames in contiguous blocks
static std::vector<int> poolDevices(std::vector<int> devices)
{
    if (devices.empty())
    {
        int n = 0;
        if (hip::load().acf_hip_device_count(&n) || n <= 0)
        {
            throw Exception(ACF_HIP_E_NODEVICE, "HipDetectorPool: no gfx950 device");
        }
        for (int i = 0; i < n; i++)
        {
            devices.push_back(i);
        }
    }
    return devices;
}

// contexts that share a device take turns with their VALU / LDS-bound kernels (acf_hip.h, option cascade_turns)
static void poolTurns(const std::vector<int>& devices, std::vector<std::unique_ptr<HipDetector>>& dets)
{
    for (size_t i = 0; i < dets.size(); i++)
    {
        if (std::count(devices.begin(), devices.end(), devices[i]) > 1)
        {
            dets[i]->setOption("cascade_turns", 5);
            dets[i]->setOption("tile_persist", 0); // (short-lived tile workgroups leave LDS for the other contexts' kernels)
            dets[i]->setOption("shared_device", 1); // (kernel forms with the least work: the other contexts cover a thin chain's latency)
        }
    }
}

HipDetectorPool::HipDetectorPool(const HipDetector::Options& o, const HipDetector::Classifier& c, std::vector<int> devices)
{
    const std::vector<int> devs = poolDevices(std::move(devices));
    for (int dev : devs)
    {
        m_dets.emplace_back(new HipDetector(o, c, dev));
    }
    poolTurns(devs, m_dets);
}

HipDetectorPool::HipDetectorPool(const std::string& filename, std::vector<int> devices)
{
    const std::vector<int> devs = poolDevices(std::move(devices));
    for (int dev : devs)
    {
        m_dets.emplace_back(new HipDetector(filename, dev));
    }
    poolTurns(devs, m_dets);
}

void HipDetectorPool::shardRange(int nFrames, int world, int rank, int& begin, int& end)
{
    const int base = nFrames / world, extra = nFrames % world;
    begin = rank * base + std::min(rank, extra);
    end = begin + base + (rank < extra ? 1 : 0);
}

int HipDetectorPool::detectBatch(const float* frames, int nFrames, int rows, int cols, int channels,
    std::vector<RectVec>& objects, std::vector<RealVec>* scores)
{
    const int world = int(m_dets.size());
    objects.assign(size_t(std::max(nFrames, 0)), RectVec());
    if (scores)
    {
        scores->assign(size_t(std::max(nFrames, 0)), RealVec());
    }
    const size_t per = size_t(rows) * cols * channels;
    std::vector<std::vector<RectVec>> obj(static_cast<size_t>(world));
    std::vector<std::vector<RealVec>> sc(static_cast<size_t>(world));
    std::vector<std::string> errs(static_cast<size_t>(world));
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
    {
        int b, e;
        shardRange(nFrames, world, r, b, e);
        if (e <= b)
        {
            continue;
        }
        th.emplace_back([&, r, b, e] {
            try
            {
                m_dets[size_t(r)]->detectBatch(frames + size_t(b) * per, e - b, rows, cols, channels, obj[size_t(r)], scores ? &sc[size_t(r)] : nullptr);
            }
            catch (const std::exception& ex)
            {
                errs[size_t(r)] = ex.what();
            }
        });
    }
    for (auto& t : th)
    {
        t.join();
    }
    for (int r = 0; r < world; r++)
    {
        if (!errs[size_t(r)].empty())
        {
            throw Exception(ACF_HIP_E_HIP, "HipDetectorPool: device " + std::to_string(r) + ": " + errs[size_t(r)]);
        }
        int b, e;
        shardRange(nFrames, world, r, b, e);
        for (int f = b; f < e; f++)
        {
            objects[size_t(f)] = std::move(obj[size_t(r)][size_t(f - b)]);
            if (scores)
            {
                (*scores)[size_t(f)] = std::move(sc[size_t(r)][size_t(f - b)]);
            }
        }
    }
    return 0;
}

} // namespace acf
