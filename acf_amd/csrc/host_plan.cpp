// host_plan.cpp — see host_plan.h.
#include "host_plan.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>

namespace acfhip
{

// Detector::getScales, chnsPyramid.cpp:461-529.  The reference works on the
// transposed image, so its sz.width is the upright height H and sz.height the
// upright width W; minDs likewise (ACFIO.h:168-181).
ScaleList getScales(int nPerOct, int nOctUp, int minDs_h, int minDs_w, int shrink, int H, int W)
{
    ScaleList out;
    if (H <= 0 || W <= 0)
    {
        return out;
    }
    const double szw = H, szh = W; // reference members
    const double ratio_w = szw / double(minDs_h), ratio_h = szh / double(minDs_w);
    // util::log2(x) = std::log(x) / std::log(2.0)  (util/acf_math.h:20-29)
    const double lg = std::log(std::min(ratio_w, ratio_h)) / std::log(2.0);
    const int nScales = int(std::floor(double(nPerOct) * (double(nOctUp) + lg) + 1.0));
    double d0 = szh, d1 = szw;
    if (szh >= szw)
    {
        std::swap(d0, d1);
    }
    std::vector<double> cand;
    for (int i = 0; i < nScales; i++)
    {
        const double s = std::pow(2.0, -double(i) / double(nPerOct) + double(nOctUp));
        const double base = std::round(d0 * s / shrink) * shrink;
        const double s0 = (base - 0.25 * shrink) / d0;
        const double s1 = (base + 0.25 * shrink) / d0;
        double bestS = 0, bestE = std::numeric_limits<double>::max();
        // the reference accumulates j += 0.01 in double; keep that exact sequence
        for (double j = 0.0; j < 1.0 - std::numeric_limits<double>::epsilon(); j += 0.01)
        {
            const double ss = (j * (s1 - s0) + s0);
            double e0 = d0 * ss;
            e0 = std::abs(e0 - std::round(e0 / shrink) * shrink);
            double e1 = d1 * ss;
            e1 = std::abs(e1 - std::round(e1 / shrink) * shrink);
            const double e = std::max(e0, e1);
            if (e < bestE)
            {
                bestS = ss;
                bestE = e;
            }
        }
        cand.push_back(bestS);
    }
    cand.push_back(0.0);
    for (size_t i = 1; i < cand.size(); i++)
    {
        if (cand[i] != cand[i - 1])
        {
            const double s = cand[i - 1];
            out.scales.push_back(s);
            out.shw_h.push_back(std::round(szw * s / shrink) * shrink / szw);
            out.shw_w.push_back(std::round(szh * s / shrink) * shrink / szh);
        }
    }
    return out;
}

// resampleCoef<float>, imResampleMex.cpp:24-121.
AxisCoef resampleCoef(int na, int nb, int pad)
{
    AxisCoef c;
    c.na = na;
    c.nb = nb;
    c.down = na > nb;
    const float s = float(nb) / float(na), sInv = 1 / s;
    const float wt0 = float(1e-3) * s;
    if (c.down)
    {
        c.start.assign(1, 0);
        for (int yb = 0; yb < nb; yb++)
        {
            const float ya0f = yb * sInv, ya1f = ya0f + sInv;
            const int ya0 = int(std::ceil(ya0f)), ya1 = int(ya1f);
            float W = 0;
            int n1 = 0;
            const size_t first = c.src.size();
            for (int ya = ya0 - 1; ya < ya1 + 1; ya++)
            {
                float wt = s;
                if (ya == ya0 - 1)
                {
                    wt = (ya0 - ya0f) * s;
                }
                else if (ya == ya1)
                {
                    wt = (ya1f - ya1) * s;
                }
                if (wt > wt0 && ya >= 0)
                {
                    c.src.push_back(ya);
                    c.wt.push_back(wt);
                    n1++;
                    W += wt;
                }
            }
            if (W > 1)
            {
                for (int i = 0; i < n1; i++)
                {
                    c.wt[first + i] /= W;
                }
            }
            c.bd[0] = std::max(c.bd[0], n1);
            while (n1 < pad)
            {
                // zero-weight filler repeating the previous source index (:84-91)
                c.src.push_back(c.src.empty() ? 0 : c.src.back());
                c.wt.push_back(0.f);
                n1++;
            }
            c.start.push_back(int(c.src.size()));
        }
    }
    else
    {
        for (int yb = 0; yb < nb; yb++)
        {
            const float yaf = (float(.5) + yb) * sInv - float(.5);
            int ya = int(std::floor(yaf));
            float wt = 1;
            if (ya >= 0 && ya < na - 1)
            {
                wt = 1 - (yaf - ya);
            }
            if (ya < 0)
            {
                ya = 0;
                c.bd[0]++;
            }
            if (ya >= na - 1)
            {
                ya = na - 1;
                c.bd[1]++;
            }
            c.src.push_back(ya);
            c.wt.push_back(wt);
        }
    }
    return c;
}

static int exactFactor(int na, int nb)
{
    for (int k = 2; k <= 4; k++)
    {
        if (na == k * nb)
        {
            return k;
        }
    }
    return 0;
}

int buildResample(int ha, int wa, int hb, int wb, ResampleDesc& d, TableArena& arena)
{
    d = ResampleDesc();
    d.ha = ha;
    d.hb = hb;
    d.wa = wa;
    d.wb = wb;
    // x axis: resampleCoef(wa, wb, pad 0) (:143)
    AxisCoef cx = resampleCoef(wa, wb, 0);
    d.xk = exactFactor(wa, wb);
    d.xbd0 = cx.bd[0];
    d.xbd1 = cx.bd[1];
    if (d.xk)
    {
        // the reference advances x1 += k and reads xas[x1] (:198-215): the first source column of group x
        d.xmode = RS_EXACT;
        d.x_src = int(arena.ints.size());
        for (int x = 0; x < wb; x++)
        {
            const size_t x1 = size_t(x) * d.xk;
            if (x1 >= cx.src.size())
            {
                return ACF_HIP_E_UNSUPPORTED;
            }
            arena.ints.push_back(cx.src[x1]);
        }
    }
    else if (cx.down)
    {
        d.xmode = RS_DOWN;
        d.x_start = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cx.start.begin(), cx.start.end());
        d.x_src = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cx.src.begin(), cx.src.end());
        d.x_wt = int(arena.floats.size());
        arena.floats.insert(arena.floats.end(), cx.wt.begin(), cx.wt.end());
    }
    else
    {
        d.xmode = RS_UP;
        d.x_src = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cx.src.begin(), cx.src.end());
        d.x_wt = int(arena.floats.size());
        arena.floats.insert(arena.floats.end(), cx.wt.begin(), cx.wt.end());
    }
    // y axis: resampleCoef(ha, hb, pad 4) (:144)
    AxisCoef cy = resampleCoef(ha, hb, 4);
    d.yk = exactFactor(ha, hb);
    d.ybd0 = cy.bd[0];
    d.ybd1 = cy.bd[1];
    if (d.yk)
    {
        d.ymode = RS_EXACT;
    }
    else if (cy.down)
    {
        if (cy.bd[0] < 2)
        {
            return ACF_HIP_E_UNSUPPORTED; // the reference leaves B unwritten in this case (:325-356)
        }
        d.ymode = RS_DOWN;
        d.y_start = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cy.start.begin(), cy.start.end());
        d.y_src = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cy.src.begin(), cy.src.end());
        d.y_wt = int(arena.floats.size());
        arena.floats.insert(arena.floats.end(), cy.wt.begin(), cy.wt.end());
    }
    else
    {
        d.ymode = RS_UP;
        d.y_src = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), cy.src.begin(), cy.src.end());
        d.y_wt = int(arena.floats.size());
        arena.floats.insert(arena.floats.end(), cy.wt.begin(), cy.wt.end());
    }
    // One 32-byte record per output column with everything the x pass of that column needs (first source
    // column, tap count, first four weights): the fused level kernel fetches it with a single scalar load
    // instead of chasing start[] -> src[] -> wt[].
    while (arena.ints.size() % 8)
    {
        arena.ints.push_back(0);
    }
    d.x_col = int(arena.ints.size());
    auto fbits = [](float f) {
        int32_t b;
        std::memcpy(&b, &f, 4);
        return b;
    };
    for (int x = 0; x < wb; x++)
    {
        int32_t rec[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (d.xmode == RS_EXACT)
        {
            rec[0] = arena.ints[size_t(d.x_src) + x];
            rec[1] = d.xk;
        }
        else if (d.xmode == RS_DOWN)
        {
            const int s0 = cx.start[size_t(x)], s1 = cx.start[size_t(x) + 1];
            rec[0] = cx.src[size_t(s0)];
            rec[1] = s1 - s0;
            rec[2] = d.x_wt + s0;
            for (int j = 0; j < 4 && j < s1 - s0; j++)
            {
                rec[4 + j] = fbits(cx.wt[size_t(s0 + j)]);
            }
        }
        else
        {
            rec[0] = cx.src[size_t(x)];
            rec[1] = 2;
            rec[3] = (x < d.xbd0 || x >= wb - d.xbd1) ? 1 : 0;
            const float w0 = cx.wt[size_t(x)];
            rec[4] = fbits(w0);
            rec[5] = fbits(1 - w0);
        }
        arena.ints.insert(arena.ints.end(), rec, rec + 8);
    }
    return ACF_HIP_OK;
}

// imResampleMex.cpp:145-157 (and :286,:309,:316 for the exact y path).
void setResampleGain(ResampleDesc& d, const double ratio[3], int c1, int c2)
{
    d.c1 = c1;
    d.c2 = c2;
    for (int j = 0; j < 3; j++)
    {
        float r = float(ratio[j]);
        if (d.wa == 2 * d.wb)
        {
            r /= 2;
        }
        if (d.wa == 3 * d.wb)
        {
            r /= 3;
        }
        if (d.wa == 4 * d.wb)
        {
            r /= 4;
        }
        r /= float(1 + 1e-6);
        d.r[j] = r;
        d.rk[j] = d.yk ? r / float(d.yk) : r;
    }
}

int colorPlanes(const acf_hip_params& p)
{
    return p.colorSpace == ACF_HIP_CS_GRAY ? 1 : 3;
}

int numChannels(const acf_hip_params& p)
{
    return (p.colorEnabled ? colorPlanes(p) : 0) + (p.gradMagEnabled ? 1 : 0) + (p.gradHistEnabled ? p.nOrients : 0);
}

// The checks on the Options::Pyramid::Chns fields (and the input's planes) that every entry computing channels makes:
// acf_hip_plan (buildPlan) and acf_hip_chns_compute.  err: the message without its "plan: " / "chns_compute: " prefix.
int checkChnsParams(const acf_hip_params& p, int d_in, std::string& err)
{
    if (p.shrink != 4 && p.shrink != 2)
    {
        err = "shrink must be 2 or 4";
        return ACF_HIP_E_UNSUPPORTED;
    }
    if (p.binSize != 0 && p.binSize != p.shrink)
    {
        err = "binSize != shrink";
        return ACF_HIP_E_UNSUPPORTED;
    }
    // gradHist's branches (gradientMex.cpp:391-509): softBin even — orientation interpolated (>= 0) or nearest bin (< 0), no spatial
    // interpolation — are built; odd softBin is the trilinear form (HOG / FHOG features, not an ACF channel set)
    if (p.softBin % 2 != 0)
    {
        err = "odd softBin (trilinear spatial binning) is not built";
        return ACF_HIP_E_UNSUPPORTED;
    }
    if (!(p.gradMagEnabled || p.gradHistEnabled || p.colorEnabled))
    {
        err = "no channels enabled";
        return ACF_HIP_E_INVALID;
    }
    if (p.colorSpace != ACF_HIP_CS_LUV && p.colorSpace != ACF_HIP_CS_GRAY && p.colorSpace != ACF_HIP_CS_ORIG && p.colorSpace != ACF_HIP_CS_RGB &&
        p.colorSpace != ACF_HIP_CS_HSV)
    {
        err = "colour space";
        return ACF_HIP_E_UNSUPPORTED;
    }
    if (p.isLuv && p.colorSpace == ACF_HIP_CS_HSV)
    {
        err = "isLuv with hsv (rgbConvert.cpp:150-155)";
        return ACF_HIP_E_INVALID;
    }
    if (d_in == 1 && !(p.colorSpace == ACF_HIP_CS_GRAY || p.colorSpace == ACF_HIP_CS_ORIG))
    {
        err = "1-plane input needs colorSpace gray or orig (rgbConvert.cpp:140-148)";
        return ACF_HIP_E_INVALID;
    }
    if (p.isLuv && p.colorSpace == ACF_HIP_CS_GRAY)
    {
        err = "isLuv with gray (rgbConvert.cpp:150-155)";
        return ACF_HIP_E_INVALID;
    }
    if (p.colorChn < 0 || p.colorChn >= colorPlanes(p))
    {
        err = "colorChn";
        return ACF_HIP_E_INVALID;
    }
    return ACF_HIP_OK;
}

int buildPlan(const acf_hip_params& p, int H, int W, int d_in, Plan& plan, std::string& err)
{
    plan = Plan();
    // d_in: 1 or 3 image planes, or 5 = three image planes + the gradient magnitude and orientation that come WITH the image
    // (chnsPyramid.cpp:248-255: the GL pipeline's LUVMO frames; M and O then replace gradientMag at the first real scale, :318-322)
    if (H <= 0 || W <= 0 || (d_in != 1 && d_in != 3 && d_in != 5))
    {
        err = "plan: bad frame geometry";
        return ACF_HIP_E_INVALID;
    }
    {
        const int rcc = checkChnsParams(p, d_in, err);
        if (rcc)
        {
            err = "plan: " + err;
            return rcc;
        }
    }
    if (p.nApprox > 0 && p.nLambdas != 3 && p.nLambdas != 0)
    {
        err = "plan: lambdas: none (estimated per image, chnsPyramid.cpp:341-374) or three (colour, gradMag, gradHist)";
        return ACF_HIP_E_INVALID;
    }
    plan.H = H;
    plan.W = W;
    plan.d_in = d_in;
    plan.d = colorPlanes(p);
    plan.nChns = numChannels(p);
    const int shrink = p.shrink;
    ScaleList sl = getScales(p.nPerOct, p.nOctUp, p.minDs_h, p.minDs_w, shrink, H, W);
    const int n = int(sl.scales.size());
    if (n == 0)
    {
        err = "plan: no scales (frame smaller than minDs)";
        return ACF_HIP_E_INVALID;
    }
    plan.levels.resize(n);
    // real/approx split and nearest real scale, chnsPyramid.cpp:272-292 (1-based there)
    std::vector<int> isR;
    for (int i = 0; i < n; i++)
    {
        if (i % (p.nApprox + 1) == 0)
        {
            isR.push_back(i + 1);
        }
    }
    std::vector<int> isH(isR.size() + 1, 0);
    isH.back() = n;
    for (int i = 0; i + 1 < int(isR.size()); i++)
    {
        isH[i + 1] = (isR[i] + isR[i + 1]) / 2;
    }
    std::vector<int> isN(n, 0);
    for (size_t i = 0; i < isR.size(); i++)
    {
        for (int j = isH[i]; j < isH[i + 1]; j++)
        {
            isN[j] = isR[i];
        }
    }
    int64_t off = 0, roff = 0;
    plan.raw_off.resize(n);
    for (int i = 0; i < n; i++)
    {
        acf_hip_level& l = plan.levels[i];
        l.scale = sl.scales[i];
        l.scalehw_h = sl.shw_h[i];
        l.scalehw_w = sl.shw_w[i];
        l.isReal = (i % (p.nApprox + 1)) == 0;
        l.realIndex = isN[i] - 1;
        l.hC = int(std::round(double(H) * l.scale / double(shrink)));
        l.wC = int(std::round(double(W) * l.scale / double(shrink)));
        l.hP = l.hC + 2 * (p.pad_h / shrink);
        l.wP = l.wC + 2 * (p.pad_w / shrink);
        l.nWinR = std::max(0, int(std::ceil(float(l.hP * shrink - p.modelDsPad_h + 1) / p.stride)));
        l.nWinC = std::max(0, int(std::ceil(float(l.wP * shrink - p.modelDsPad_w + 1) / p.stride)));
        l.offset = off;
        off += int64_t(plan.nChns) * l.hP * l.wP;
        plan.raw_off[i] = roff;
        roff += int64_t(plan.nChns) * l.hC * l.wC;
        if (l.hC < 4 || l.wC < 4)
        {
            err = "plan: level smaller than 4 cells (convTri's sepFilter2D fallback, convTri.cpp:224-251, is not implemented)";
            return ACF_HIP_E_UNSUPPORTED;
        }
        if (l.isReal)
        {
            plan.real.push_back(i);
            plan.real_h.push_back(l.hC * shrink);
            plan.real_w.push_back(l.wC * shrink);
            const int m = std::min(l.hC, l.wC) * shrink;
            if (p.normRad != 0 && (2 * p.normRad + 1) >= m)
            {
                err = "plan: normRad too large for the smallest real scale";
                return ACF_HIP_E_UNSUPPORTED;
            }
        }
    }
    plan.pyr_floats = off;
    plan.raw_floats = roff;
    if (d_in == 5 && (plan.real.empty() || plan.real_h[0] != H || plan.real_w[0] != W))
    {
        // (the reference pushes the full-size M, O planes onto the first real scale's image whatever its size, chnsPyramid.cpp:318-322:
        // only a first real scale of the image's own size makes sense of that)
        err = "plan: five input planes (image + M, O) need the first real scale to be the image's own size (nOctUp = 0, sizes divisible by shrink)";
        return ACF_HIP_E_UNSUPPORTED;
    }
    plan.lambdaLevel[0] = plan.lambdaLevel[1] = -1;
    if (p.nApprox > 0 && p.nLambdas == 0)
    {
        // image-specific lambdas: the two real levels they are estimated from (chnsPyramid.cpp:343-355, 1-based there)
        std::vector<int> is;
        for (int i = 1 + p.nOctUp * p.nPerOct; i <= n; i += p.nApprox + 1)
        {
            is.push_back(i - 1);
        }
        if (is.size() < 2)
        {
            err = "plan: image-specific lambdas need two real scales at or below the image size (CV_Assert(is.size() >= 2), chnsPyramid.cpp:351)";
            return ACF_HIP_E_INVALID;
        }
        plan.lambdaLevel[0] = is.size() > 2 ? is[1] : is[0];
        plan.lambdaLevel[1] = is.size() > 2 ? is[2] : is[1];
    }
    return ACF_HIP_OK;
}


// ---- threshold-rank cells (host_plan.h) ------------------------------------------------------------------------------
namespace
{
inline int32_t rankKey(float v)
{
    int32_t b;
    std::memcpy(&b, &v, 4);
    return b > 0 ? b : 0;
}
inline int rankBucket(const RankChan& c, int32_t key)
{
    const int b = (key >> c.shift) - c.base;
    return b < 0 ? 0 : (b > c.nb - 1 ? c.nb - 1 : b);
}
} // namespace

uint32_t RankTables::rankOfCell(int chn, float v) const
{
    const RankChan& c = chan[size_t(chn)];
    if (v < 0.f)
    {
        return 0;
    }
    const int32_t key = rankKey(v);
    const RankRec& r = rec[size_t(c.recOff) + size_t(rankBucket(c, key))];
    const uint32_t low = uint32_t(key) & ((1u << c.shift) - 1u);
    uint32_t n = r.lo;
    for (int j = 0; j < RANK_WINDOW; j++)
    {
        n += (uint32_t(r.t(j)) <= low) ? 1u : 0u;
    }
    return n;
}

uint32_t RankTables::rankOfThreshold(int chn, float t) const
{
    const RankChan& c = chan[size_t(chn)];
    if (!(t == t))
    {
        return 0; // `ftr < NaN` is never true
    }
    const float* b = thr.data() + c.thrOff;
    return uint32_t(std::lower_bound(b, b + c.nThr, t) - b) + 1u; // t is one of the channel's thresholds: index + 1
}

void buildRankTables(const float* thrs, const int32_t* chnOfNode, size_t nNodes, int nChns, RankTables& out)
{
    out = RankTables();
    out.chan.assign(size_t(nChns), RankChan{});
    std::vector<std::vector<float>> per(static_cast<size_t>(nChns));
    for (size_t q = 0; q < nNodes; q++)
    {
        const int z = chnOfNode[q];
        if (z < 0)
        {
            continue;
        }
        if (z >= nChns)
        {
            out.why = "feature channel out of range";
            return;
        }
        const float t = thrs[q];
        if (t < 0.f)
        {
            out.why = "negative threshold (rank cells place every negative cell below all thresholds)";
            return;
        }
        if (t == t) // NaN thresholds have no rank (never true)
        {
            per[size_t(z)].push_back(t == 0.f ? 0.f : t); // -0.0 and +0.0 are one threshold
        }
    }
    for (int z = 0; z < nChns; z++)
    {
        std::vector<float>& t = per[size_t(z)];
        std::sort(t.begin(), t.end());
        t.erase(std::unique(t.begin(), t.end()), t.end());
        RankChan& c = out.chan[size_t(z)];
        c.nThr = int32_t(t.size());
        if (t.size() > 65534)
        {
            out.why = "more than 65534 distinct thresholds in one channel";
            return;
        }
        // a threshold of exactly 0 is <= every v >= -0 (and a negative v ranks 0 anyway): it only shifts every record's
        // `lo` by one, and stays out of the bucket geometry (its key, 0, is far below every positive float's)
        std::vector<float> tAll = t;
        const uint32_t hasZero = (!t.empty() && t.front() == 0.f) ? 1u : 0u;
        if (hasZero)
        {
            t.erase(t.begin());
        }
        // the largest shift (smallest table) whose buckets hold at most RANK_WINDOW thresholds each
        int best = -1;
        for (int shift = 15; shift >= 0 && best < 0; shift--)
        {
            if (t.empty())
            {
                best = shift;
                break;
            }
            const int64_t kLo = rankKey(t.front()) >> shift, kHi = rankKey(t.back()) >> shift;
            if (kHi - kLo + 3 > RANK_MAX_BUCKETS)
            {
                break; // finer buckets only get more numerous
            }
            int run = 0, worst = 0;
            int64_t prev = -1;
            for (float v : t)
            {
                const int64_t k = rankKey(v) >> shift;
                run = (k == prev) ? run + 1 : 1;
                prev = k;
                worst = std::max(worst, run);
            }
            if (worst <= RANK_WINDOW)
            {
                best = shift;
            }
        }
        if (best < 0)
        {
            out.why = "thresholds of a channel spread too wide or packed too densely for a bucket table of RANK_MAX_BUCKETS records";
            return;
        }
        c.shift = best;
        c.base = t.empty() ? 0 : int32_t((rankKey(t.front()) >> best) - 1);
        c.nb = t.empty() ? 1 : int32_t((rankKey(t.back()) >> best) - c.base + 2);
        c.recOff = int32_t(out.rec.size());
        c.thrOff = int32_t(out.thr.size());
        RankRec empty;
        empty.lo = 0;
        for (int j = 0; j < RANK_WINDOW; j++)
        {
            empty.t(j) = RANK_UNUSED;
        }
        out.rec.resize(out.rec.size() + size_t(c.nb), empty);
        RankRec* rec = out.rec.data() + c.recOff;
        std::vector<int> fill(static_cast<size_t>(c.nb), 0);
        for (size_t j = 0; j < t.size(); j++)
        {
            const int32_t key = rankKey(t[j]);
            const int b = rankBucket(c, key);
            rec[b].t(fill[size_t(b)]++) = uint16_t(uint32_t(key) & ((1u << best) - 1u));
        }
        uint32_t acc = hasZero;
        for (int b = 0; b < c.nb; b++)
        {
            rec[b].lo = uint16_t(acc); // thresholds in lower buckets
            acc += uint32_t(fill[size_t(b)]);
        }
        out.thr.insert(out.thr.end(), tAll.begin(), tAll.end());
        out.maxRec = std::max(out.maxRec, c.nb);
    }
    out.ok = true;
}

} // namespace acfhip


// ---- resize to a minimum object width (apps' Resizer; OpenCV's CV_8U resize restated)
namespace acfhip
{
namespace
{
int cvRoundD(double v) { return int(std::lrint(v)); }
int cvFloorD(double v)
{
    const int i = int(v);
    return i - (i > v);
}
int cvCeilD(double v)
{
    const int i = int(v);
    return i + (i < v);
}
int satS16(float v)
{
    const long i = std::lrintf(v);
    return int(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}
void areaTab(int ssize, int dsize, double scale, std::vector<int32_t>& run, std::vector<int32_t>& tap)
{
    run.assign(size_t(dsize) * 2, 0);
    tap.clear();
    auto push = [&](int dx, int si, float alpha) {
        int32_t bits;
        std::memcpy(&bits, &alpha, 4);
        if (run[size_t(dx) * 2 + 1] == 0)
        {
            run[size_t(dx) * 2] = int32_t(tap.size() / 2);
        }
        run[size_t(dx) * 2 + 1]++;
        tap.push_back(si);
        tap.push_back(bits);
    };
    for (int dx = 0; dx < dsize; dx++)
    {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = cvCeilD(fsx1), sx2 = cvFloorD(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3)
        {
            push(dx, sx1 - 1, float((sx1 - fsx1) / cellWidth));
        }
        for (int sx = sx1; sx < sx2; sx++)
        {
            push(dx, sx, float(1.0 / cellWidth));
        }
        if (fsx2 - sx2 > 1e-3)
        {
            push(dx, sx2, float(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth));
        }
    }
}
} // namespace

void resizeDims(int rows, int cols, double scale, int& drows, int& dcols)
{
    dcols = cvRoundD(cols * scale);
    drows = cvRoundD(rows * scale);
}

int buildResizeTables(int rows, int cols, double scale, ResizeTables& t)
{
    t = ResizeTables();
    if (rows < 1 || cols < 1 || !(scale > 0))
    {
        return 1;
    }
    t.rows = rows;
    t.cols = cols;
    resizeDims(rows, cols, scale, t.drows, t.dcols);
    if (t.drows < 1 || t.dcols < 1)
    {
        return 1;
    }
    const double sc = 1. / scale; // scale_x == scale_y: the caller's factor, not recomputed from the sizes
    const int isc = cvRoundD(sc);
    const bool areaFast = std::fabs(sc - isc) < 2.220446049250313e-16;
    // Resizer: INTER_AREA when scale < 1, INTER_LINEAR otherwise; cv::resize takes the area form only when reducing (and takes it
    // for INTER_LINEAR at exactly 1/2)
    if (scale < 1.f && sc >= 1)
    {
        if (areaFast)
        {
            t.mode = 2;
            t.isx = t.isy = isc;
            return 0;
        }
        t.mode = 1;
        areaTab(cols, t.dcols, sc, t.xrun, t.xtap);
        areaTab(rows, t.drows, sc, t.yrun, t.ytap);
        return 0;
    }
    t.mode = 0;
    t.xlin.resize(size_t(t.dcols) * 4);
    t.ylin.resize(size_t(t.drows) * 4);
    int xmax = t.dcols;
    for (int dx = 0; dx < t.dcols; dx++)
    {
        float f = float((dx + 0.5) * sc - 0.5);
        int sx = cvFloorD(f);
        f -= sx;
        if (sx < 0)
        {
            f = 0, sx = 0;
        }
        if (sx + 1 >= cols)
        {
            xmax = std::min(xmax, dx);
            if (sx >= cols - 1)
            {
                f = 0, sx = cols - 1;
            }
        }
        t.xlin[size_t(dx) * 4] = sx;
        t.xlin[size_t(dx) * 4 + 1] = satS16((1.f - f) * 2048.f);
        t.xlin[size_t(dx) * 4 + 2] = satS16(f * 2048.f);
    }
    for (int dx = 0; dx < t.dcols; dx++)
    {
        t.xlin[size_t(dx) * 4 + 3] = dx < xmax ? 1 : 0;
    }
    for (int dy = 0; dy < t.drows; dy++)
    {
        float f = float((dy + 0.5) * sc - 0.5);
        const int sy = cvFloorD(f);
        f -= sy;
        t.ylin[size_t(dy) * 4] = std::min(std::max(sy, 0), rows - 1);
        t.ylin[size_t(dy) * 4 + 1] = std::min(std::max(sy + 1, 0), rows - 1);
        t.ylin[size_t(dy) * 4 + 2] = satS16((1.f - f) * 2048.f);
        t.ylin[size_t(dy) * 4 + 3] = satS16(f * 2048.f);
    }
    return 0;
}
} // namespace acfhip
