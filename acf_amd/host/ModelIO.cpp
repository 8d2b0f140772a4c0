#include "ModelIO.h"

#include <cstring>
#include <fstream>
#include <istream>
#include <map>
#include <ostream>
#include <set>
#include <sstream>

namespace acf
{

namespace
{

// One description of the layout drives both directions, the way one cereal serialize() does.
class Archive
{
public:
    explicit Archive(std::istream& is) : m_in(&is)
    {
        uint8_t flag = 0;
        raw(&flag, 1);
        if (flag > 1)
        {
            bad("not a PortableBinary stream (endianness flag)");
        }
        m_swap = (flag == 0); // stream is big-endian; this host (x86-64) is little-endian
    }
    explicit Archive(std::ostream& os) : m_out(&os)
    {
        uint8_t flag = 1;
        raw(&flag, 1);
    }
    bool loading() const { return m_in != nullptr; }

    template <class T>
    void pod(T& v)
    {
        raw(&v, sizeof(T));
        if (m_swap && loading())
        {
            unsigned char* b = reinterpret_cast<unsigned char*>(&v);
            for (size_t i = 0; i < sizeof(T) / 2; i++)
            {
                std::swap(b[i], b[sizeof(T) - 1 - i]);
            }
        }
    }
    void flag(bool& v)
    {
        uint8_t b = v ? 1 : 0;
        raw(&b, 1);
        v = b != 0;
    }
    void size(uint64_t& n)
    {
        pod(n);
        if (loading() && n > (uint64_t(1) << 31))
        {
            bad("implausible container size");
        }
    }
    void str(std::string& s)
    {
        uint64_t n = s.size();
        size(n);
        s.resize(size_t(n));
        raw(n ? &s[0] : nullptr, size_t(n));
    }
    template <class T>
    void vec(std::vector<T>& v)
    {
        uint64_t n = v.size();
        size(n);
        v.resize(size_t(n));
        for (auto& e : v)
        {
            pod(e);
        }
    }
    // cereal writes a class version the first time a type is met (registerClassVersion / loadClassVersion)
    void version(const char* type, uint32_t ver = 0)
    {
        if (m_seen.insert(type).second)
        {
            pod(ver);
        }
    }
    void raw(void* p, size_t n)
    {
        if (n == 0)
        {
            return;
        }
        if (m_in)
        {
            m_in->read(static_cast<char*>(p), std::streamsize(n));
            if (!*m_in)
            {
                bad("unexpected end of stream");
            }
        }
        else
        {
            m_out->write(static_cast<const char*>(p), std::streamsize(n));
        }
    }
    [[noreturn]] static void bad(const char* what) { throw Exception(ACF_HIP_E_INVALID, std::string("cpb: ") + what); }

private:
    std::istream* m_in = nullptr;
    std::ostream* m_out = nullptr;
    bool m_swap = false;
    std::set<std::string> m_seen;
};

// Field<T> := ver, value, name, has, isLeaf (ACFField.h:123-130).  `value` is touched only when the stored
// field says `has` (the reference leaves a default-constructed value otherwise).
template <class T, class Fn>
void field(Archive& ar, const char* type, const char* name, T& value, bool isLeaf, Fn&& body)
{
    ar.version(type);
    T tmp = value;
    body(tmp);
    std::string nm = name;
    bool has = true, leaf = isLeaf;
    ar.str(nm);
    ar.flag(has);
    ar.flag(leaf);
    if (ar.loading() && nm != name)
    {
        // the layout is restated from the reference's serialisers, not checked against a cereal-written file: a drift in
        // field order or count must fail here, not misparse a model silently
        Archive::bad((std::string("field name mismatch: expected '") + name + "', found '" + nm + "'").c_str());
    }
    if (!ar.loading() || has)
    {
        value = tmp;
    }
}
// name / has / isLeaf trailer of a Field<struct>; on load the stored name must be the expected one
void checkedName(Archive& ar, std::string& nm, bool& has, bool& leaf)
{
    const std::string want = nm;
    ar.str(nm), ar.flag(has), ar.flag(leaf);
    if (ar.loading() && nm != want)
    {
        Archive::bad(("field name mismatch: expected '" + want + "', found '" + nm + "'").c_str());
    }
}
void fInt(Archive& ar, const char* name, int& v)
{
    field(ar, "Field<int>", name, v, true, [&](int& x) { int32_t t = x; ar.pod(t); x = t; });
}
void fDouble(Archive& ar, const char* name, double& v)
{
    field(ar, "Field<double>", name, v, true, [&](double& x) { ar.pod(x); });
}
void fString(Archive& ar, const char* name, std::string& v)
{
    field(ar, "Field<string>", name, v, true, [&](std::string& x) { ar.str(x); });
}
void fSize(Archive& ar, const char* name, Size& v)
{
    field(ar, "Field<Size>", name, v, true, [&](Size& x) {
        ar.version("cv::Size"); // ACFIOArchive.h:48-53
        int32_t w = x.width, h = x.height;
        ar.pod(w);
        ar.pod(h);
        x.width = w;
        x.height = h;
    });
}
void fVecDouble(Archive& ar, const char* name, std::vector<double>& v)
{
    field(ar, "Field<vector<double>>", name, v, true, [&](std::vector<double>& x) { ar.vec(x); });
}
void fVecInt(Archive& ar, const char* name, std::vector<int32_t>& v)
{
    field(ar, "Field<vector<int>>", name, v, true, [&](std::vector<int32_t>& x) { ar.vec(x); });
}

// cv::Mat (io/cvmat_cereal.h:20-73); CV_32S = 4, CV_32F = 5.  `ignored`: a matrix the hot path never reads
// (weights, depth): any element type is accepted and the payload skipped.
template <class T>
void mat(Archive& ar, int cvType, int& rows, int& cols, std::vector<T>& data, bool ignored = false)
{
    ar.version("cv::Mat");
    int32_t r = rows, c = cols, t = cvType;
    bool continuous = true;
    ar.pod(r);
    ar.pod(c);
    ar.pod(t);
    ar.flag(continuous);
    if (ar.loading())
    {
        if (r < 0 || c < 0 || int64_t(r) * c > (int64_t(1) << 28))
        {
            Archive::bad("implausible matrix size");
        }
        const size_t n = size_t(r) * size_t(c);
        if (ignored)
        {
            static const int depthBytes[8] = { 1, 1, 2, 2, 4, 4, 8, 2 };
            const size_t bytes = n * size_t(depthBytes[t & 7]) * size_t(((t >> 3) & 511) + 1);
            std::vector<char> skip(bytes);
            ar.raw(skip.data(), bytes);
            return;
        }
        if (n > 0 && t != cvType)
        {
            Archive::bad("unexpected matrix element type");
        }
        rows = r;
        cols = c;
        data.resize(n);
    }
    // continuous: one block; otherwise row by row — the same bytes in the same order
    for (auto& e : data)
    {
        ar.pod(e);
    }
}

void serializeClassifier(Archive& ar, HipDetector::Classifier& c)
{
    ar.version("Classifier");
    int rows = c.nTrees, cols = c.nTreeNodes;
    std::vector<int32_t> fids(c.fids.begin(), c.fids.end()), child(c.child.begin(), c.child.end());
    mat(ar, 4, rows, cols, fids);
    const int nTrees = rows, nNodes = cols;
    mat(ar, 5, rows, cols, c.thrs);
    mat(ar, 4, rows, cols, child);
    mat(ar, 5, rows, cols, c.hs);
    // weights / depth are training by-products the hot path never reads; written as the zero / level tables
    std::vector<float> weights(ar.loading() ? 0 : fids.size(), 0.f);
    std::vector<int32_t> depth;
    if (!ar.loading())
    {
        depth.resize(fids.size());
        for (size_t i = 0; i < depth.size(); i++)
        {
            int k = int(i % size_t(std::max(nNodes, 1))), d = 0;
            while (k > 0)
            {
                k = (k - 1) / 2;
                d++;
            }
            depth[i] = d;
        }
    }
    int wr = nTrees, wc = nNodes;
    mat(ar, 5, wr, wc, weights, true);
    wr = nTrees, wc = nNodes;
    mat(ar, 4, wr, wc, depth, true);
    std::vector<double> errs, losses;
    ar.vec(errs);
    ar.vec(losses);
    int32_t td = c.treeDepth;
    ar.pod(td);
    if (ar.loading())
    {
        c.nTrees = nTrees;
        c.nTreeNodes = nNodes;
        c.treeDepth = td;
        c.fids.assign(fids.begin(), fids.end());
        c.child.assign(child.begin(), child.end());
        c.thrsU8.clear();
        const size_t n = size_t(nTrees) * size_t(nNodes);
        if (c.thrs.size() != n || c.hs.size() != n || (c.child.size() != n && !c.child.empty()))
        {
            Archive::bad("classifier matrices disagree in size");
        }
    }
}

void serializeOptions(Archive& ar, HipDetector::Options& o)
{
    ar.version("Options");
    int zero = 0, one = 1;
    double dzero = 0;
    std::string empty;
    // ---- pPyramid : Field<Pyramid>
    ar.version("Field<Pyramid>");
    {
        auto& p = o.pPyramid;
        ar.version("Pyramid"); // ACFIOArchive.h:150-163
        ar.version("Field<Chns>");
        {
            auto& ch = p.pChns;
            ar.version("Chns"); // :174-184
            fInt(ar, "shrink", ch.shrink);
            int complete = 1;
            fInt(ar, "complete", complete);
            ar.version("Field<Color>");
            {
                ar.version("Color"); // :186-192
                fInt(ar, "enabled", ch.pColor.enabled);
                fDouble(ar, "smooth", ch.pColor.smooth);
                fString(ar, "colorSpace", ch.pColor.colorSpace);
            }
            std::string nm = "pColor";
            bool has = true, leaf = false;
            checkedName(ar, nm, has, leaf);
            ar.version("Field<GradMag>");
            {
                ar.version("GradMag"); // :194-202
                fInt(ar, "enabled", ch.pGradMag.enabled);
                fInt(ar, "colorChn", ch.pGradMag.colorChn);
                fInt(ar, "normRad", ch.pGradMag.normRad);
                fDouble(ar, "normConst", ch.pGradMag.normConst);
                fInt(ar, "full", ch.pGradMag.full);
            }
            nm = "pGradMag", has = true, leaf = false;
            checkedName(ar, nm, has, leaf);
            ar.version("Field<GradHist>");
            {
                ar.version("GradHist"); // :204-214
                fInt(ar, "enabled", ch.pGradHist.enabled);
                fInt(ar, "binSize", ch.pGradHist.binSize);
                fInt(ar, "nOrients", ch.pGradHist.nOrients);
                fInt(ar, "softBin", ch.pGradHist.softBin);
                int useHog = 0;
                double clipHog = 0.2;
                fInt(ar, "useHog", useHog);
                fDouble(ar, "clipHog", clipHog);
            }
            nm = "pGradHist", has = true, leaf = false;
            checkedName(ar, nm, has, leaf);
        }
        std::string nm = "pChns";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
        fInt(ar, "nPerOct", p.nPerOct);
        fInt(ar, "nOctUp", p.nOctUp);
        fInt(ar, "nApprox", p.nApprox);
        fVecDouble(ar, "lambdas", p.lambdas);
        fSize(ar, "pad", p.pad);
        fSize(ar, "minDs", p.minDs);
        fDouble(ar, "smooth", p.smooth);
        int concat = 1, complete = 1;
        fInt(ar, "concat", concat);
        fInt(ar, "complete", complete);
    }
    {
        std::string nm = "pPyramid";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
    }
    fSize(ar, "modelDs", o.modelDs);
    fSize(ar, "modelDsPad", o.modelDsPad);
    // ---- pNms : Field<Nms> (:165-171)
    ar.version("Field<Nms>");
    {
        ar.version("Nms");
        fString(ar, "type", o.pNms.type);
        fDouble(ar, "overlap", o.pNms.overlap);
        fString(ar, "ovrDnm", o.pNms.ovrDnm);
    }
    {
        std::string nm = "pNms";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
    }
    fInt(ar, "stride", o.stride);
    fDouble(ar, "cascThr", o.cascThr);
    fDouble(ar, "cascCal", o.cascCal);
    std::vector<int32_t> nWeak;
    fVecInt(ar, "nWeak", nWeak);
    // ---- pBoost : Field<Boost> (:131-148) — training only, read and dropped
    ar.version("Field<Boost>");
    {
        ar.version("Boost");
        ar.version("Field<Tree>");
        {
            ar.version("Tree");
            int nBins = 256, maxDepth = 2, nThreads = 16;
            double minWeight = 0.01, fracFtrs = 1;
            fInt(ar, "nBins", nBins);
            fInt(ar, "maxDepth", maxDepth);
            fDouble(ar, "minWeight", minWeight);
            fDouble(ar, "fracFtrs", fracFtrs);
            fInt(ar, "nThreads", nThreads);
        }
        std::string nm = "pTree";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
        int nW = 128, discrete = 1, verbose = 16;
        fInt(ar, "nWeak", nW);
        fInt(ar, "discrete", discrete);
        fInt(ar, "verbose", verbose);
    }
    {
        std::string nm = "pBoost";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
    }
    // ---- training bookkeeping (:115-128)
    for (const char* k : { "posGtDir", "posImgDir", "negImgDir", "posWinDir", "negWinDir" })
    {
        std::string s = empty;
        fString(ar, k, s);
    }
    for (const char* k : { "nPos", "nNeg", "nPerNeg", "nAccNeg" })
    {
        int v = zero;
        fInt(ar, k, v);
    }
    ar.version("Field<Jitter>");
    {
        ar.version("Jitter");
        int flip = zero;
        fInt(ar, "flip", flip);
    }
    {
        std::string nm = "pJitter";
        bool has = true, leaf = false;
        checkedName(ar, nm, has, leaf);
    }
    int winsSave = zero;
    fInt(ar, "winsSave", winsSave);
    (void)one;
    (void)dzero;
}

void serializeDetector(Archive& ar, HipDetector::Options& o, HipDetector::Classifier& c)
{
    ar.version("Detector", 1); // CEREAL_CLASS_VERSION(acf::Detector, 1), ACFIOArchiveCereal.cpp:7
    serializeClassifier(ar, c); // ar & clf
    serializeOptions(ar, o);    // ar & opts
}

} // namespace

void loadCpb(std::istream& is, HipDetector::Options& opts, HipDetector::Classifier& clf)
{
    Archive ar(is);
    serializeDetector(ar, opts, clf);
    // the archive holds exactly one Detector: trailing bytes mean the layout above is not the file's
    if (is.peek() != std::istream::traits_type::eof())
    {
        Archive::bad("trailing bytes after the detector");
    }
    if (clf.nTrees < 0 || clf.nTreeNodes < 0 || clf.nTrees > (1 << 20) || clf.nTreeNodes > (1 << 12))
    {
        Archive::bad("implausible classifier size");
    }
}

void saveCpb(std::ostream& os, const HipDetector::Options& opts, const HipDetector::Classifier& clf)
{
    Archive ar(os);
    HipDetector::Options o = opts;
    HipDetector::Classifier c = clf;
    serializeDetector(ar, o, c);
}

bool loadAcfm(std::istream& is, HipDetector::Options& o, HipDetector::Classifier& c)
{
    std::string line;
    if (!std::getline(is, line) || line != "ACFHIPM1")
    {
        return false;
    }
    std::map<std::string, std::string> kv;
    while (std::getline(is, line) && line != "END")
    {
        const size_t sp = line.find(' ');
        if (sp != std::string::npos)
        {
            kv[line.substr(0, sp)] = line.substr(sp + 1);
        }
    }
    auto I = [&](const char* k) { return std::stoi(kv.at(k)); };
    auto D = [&](const char* k) { return std::stod(kv.at(k)); };
    c.nTrees = I("nTrees");
    c.nTreeNodes = I("nTreeNodes");
    c.treeDepth = I("treeDepth");
    o.modelDs = Size(I("modelDs_h"), I("modelDs_w")); // {width = image-height axis}
    o.modelDsPad = Size(I("modelDsPad_h"), I("modelDsPad_w"));
    o.stride = I("stride");
    o.cascThr = D("cascThr");
    auto& p = o.pPyramid;
    p.nPerOct = I("nPerOct");
    p.nOctUp = I("nOctUp");
    p.nApprox = I("nApprox");
    p.lambdas.clear();
    {
        std::istringstream ls(kv["lambdas"]);
        double v;
        while (ls >> v)
        {
            p.lambdas.push_back(v);
        }
    }
    p.pad = Size(I("pad_h"), I("pad_w"));
    p.minDs = Size(I("minDs_h"), I("minDs_w"));
    p.smooth = D("smooth");
    p.pChns.shrink = I("shrink");
    p.pChns.pColor.enabled = I("colorEnabled");
    p.pChns.pColor.smooth = D("colorSmooth");
    const char* cs[] = { "gray", "rgb", "luv", "hsv", "orig" };
    p.pChns.pColor.colorSpace = cs[I("colorSpace")];
    p.pChns.pGradMag.enabled = I("gradMagEnabled");
    p.pChns.pGradMag.colorChn = I("colorChn");
    p.pChns.pGradMag.normRad = I("normRad");
    p.pChns.pGradMag.normConst = D("normConst");
    p.pChns.pGradMag.full = I("full");
    p.pChns.pGradHist.enabled = I("gradHistEnabled");
    p.pChns.pGradHist.binSize = I("binSize");
    p.pChns.pGradHist.nOrients = I("nOrients");
    p.pChns.pGradHist.softBin = I("softBin");
    if (c.nTrees < 0 || c.nTreeNodes < 0 || c.nTrees > (1 << 20) || c.nTreeNodes > (1 << 12))
    {
        return false; // header values are untrusted: bound them before sizing the arrays
    }
    const size_t n = size_t(c.nTrees) * c.nTreeNodes;
    c.fids.resize(n);
    c.thrs.resize(n);
    c.hs.resize(n);
    c.child.resize(n);
    is.read(reinterpret_cast<char*>(c.fids.data()), std::streamsize(n * 4));
    is.read(reinterpret_cast<char*>(c.thrs.data()), std::streamsize(n * 4));
    is.read(reinterpret_cast<char*>(c.hs.data()), std::streamsize(n * 4));
    is.read(reinterpret_cast<char*>(c.child.data()), std::streamsize(n * 4));
    o.ldcfK = 0;
    o.ldcfFilters.clear();
    if (kv.count("ldcfK") && kv.count("ldcfCount"))
    {
        o.ldcfK = I("ldcfK");
        const unsigned long cnt = std::stoul(kv.at("ldcfCount"));
        if (o.ldcfK < 0 || o.ldcfK > 16 || cnt > (1ul << 24))
        {
            return false;
        }
        o.ldcfFilters.resize(size_t(cnt));
        is.read(reinterpret_cast<char*>(o.ldcfFilters.data()), std::streamsize(o.ldcfFilters.size() * 4));
    }
    return bool(is);
}

int loadModelAny(const std::string& filename, HipDetector::Options& opts, HipDetector::Classifier& clf)
{
    std::ifstream is(filename, std::ios::binary);
    if (!is)
    {
        return -1;
    }
    if (filename.find(".cpb") != std::string::npos) // ACFIO.cpp:204
    {
        loadCpb(is, opts, clf);
        return 0;
    }
    return loadAcfm(is, opts, clf) ? 0 : -1;
}

} // namespace acf
