// host_plan.h — host-side planning for the HIP pyramid: scale list, level
// geometry and resampling coefficient tables.  Pure C++ (no HIP), so it also
// builds and is unit-tested on a machine without a GPU.
//
// Follows Detector::getScales / chnsPyramid's scale bookkeeping
// (reference chnsPyramid.cpp:270-292,461-529) and resampleCoef
// (toolbox/imResampleMex.cpp:24-121).  Shares no code with the CPU checker.
#pragma once

#include "../../include/acf_hip.h"

#include <cstdint>
#include <string>
#include <vector>

namespace acfhip
{

struct ScaleList
{
    std::vector<double> scales, shw_h, shw_w;
};

// chnsPyramid.cpp:461-529 in upright terms (H = image height, W = width).
ScaleList getScales(int nPerOct, int nOctUp, int minDs_h, int minDs_w, int shrink, int H, int W);

// One axis of imResample's coefficient set (imResampleMex.cpp:24-121).
struct AxisCoef
{
    int na = 0, nb = 0;
    bool down = false;     // na > nb
    int bd[2] = { 0, 0 };  // down: bd[0] = max taps per output; up: clamped counts at both borders
    // One entry per tap.  down: entries grouped per output (start[nb+1]); `pad`
    // zero-weight entries appended per output as the reference does.  up: one
    // entry per output.
    std::vector<int> start;
    std::vector<int> src;
    std::vector<float> wt;
};

AxisCoef resampleCoef(int na, int nb, int pad);

// Device-facing description of one resample (all planes of one level/type set).
enum
{
    RS_EXACT = 0, // na == k*nb, k in {2,3,4}
    RS_DOWN = 1,
    RS_UP = 2
};

struct ResampleDesc
{
    int32_t ha, hb, wa, wb;
    int32_t xmode, ymode;
    int32_t xk, yk;
    int32_t xbd0, xbd1; // up: border counts; down: max taps
    int32_t ybd0, ybd1;
    int32_t x_start, x_src, x_wt; // offsets into the int / float table arenas
    int32_t y_start, y_src, y_wt;
    int32_t c1, c2;    // planes [0,c1) use r[0], [c1,c2) r[1], rest r[2]
    float r[3];        // gain after the reference's /k and /(1+1e-6) (imResampleMex.cpp:145-157)
    float rk[3];       // r / yk for the exact y path (:286,:309,:316)
    int32_t nplanes;
    int32_t x_col;     // int-arena offset (multiple of 8) of the per-output-column records {xa, m, wofs, border, w0..w3 bits}
    int64_t src_off, dst_off; // float offsets inside the per-frame source / destination buffers
    int64_t src_frame_stride, dst_frame_stride;
};

struct TableArena
{
    std::vector<int32_t> ints;
    std::vector<float> floats;
};

// Build the descriptor + tables for resampling (ha,wa) -> (hb,wb).  Returns
// ACF_HIP_OK or ACF_HIP_E_UNSUPPORTED (degenerate coefficient sets the
// reference itself mishandles).
int buildResample(int ha, int wa, int hb, int wb, ResampleDesc& d, TableArena& arena);
void setResampleGain(ResampleDesc& d, const double ratio[3], int c1, int c2);

struct Plan
{
    int H = 0, W = 0, d_in = 0, d = 0; // d = colour planes after rgbConvert
    int nChns = 0;
    std::vector<acf_hip_level> levels;
    std::vector<int> real;          // level indices of real scales, in order
    std::vector<int> real_h, real_w; // image size at each real scale (multiples of shrink)
    int64_t pyr_floats = 0;          // padded, fused (what the cascade reads)
    int64_t raw_floats = 0;          // unpadded, unsmoothed channels
    std::vector<int64_t> raw_off;    // per level
    int lambdaLevel[2] = { -1, -1 }; // image-specific lambdas (no lambdas in the model): the two real levels they come from
};

int buildPlan(const acf_hip_params& p, int H, int W, int d_in, Plan& plan, std::string& err);
int checkChnsParams(const acf_hip_params& p, int d_in, std::string& err);

// ---- threshold-rank cells (the cascade's 16-bit pyramid) --------------------------------------------------------------
// The cascade only ever evaluates `chns[cid] < thrs[node]` (acfDetect1.cpp:102-104,157-166).  Per channel, let
// t_0 < t_1 < ... < t_{m-1} be the distinct thresholds of the model's nodes whose feature lies in that channel, and
//     rank(v) = #{ j : t_j <= v }            (0 .. m, a 16-bit value for m <= 65534).
// Then for every cell value v and every node threshold t_k:   v < t_k  <=>  rank(v) <= k  <=>  rank(v) < k + 1,
// so a pyramid of rank cells compared against `k + 1` gives the cascade exactly the decisions of the float pyramid: same
// leaves, same sums, same hits, in half the bytes.
//
// rank(v) without a search.  For v >= 0 the bit pattern of v is an order-preserving integer key.  Its top bits select a
// bucket; a bucket's 16-byte record holds `lo` = the number of thresholds in lower buckets and the LOW `shift` bits of the
// keys of its own thresholds (at most RANK_WINDOW = 7 of them; 0x8000 in unused slots).  Inside a bucket all keys share
// their top bits, so comparing low bits compares the floats:
//     key = max(int(bits(v)), 0);  b = clamp((key >> shift) - base, 0, nb - 1);  low = key & ((1 << shift) - 1);
//     rec = table[b] (one 16-byte read);  rank = rec.lo + #{ j : rec.t[j] <= low };   v < 0: rank = 0.
// buildRankTables picks each channel's `shift` (<= 15, so that 0x8000 is above every low key) as the largest — i.e. the
// smallest table — whose buckets hold at most RANK_WINDOW thresholds, or reports that none exists within
// RANK_MAX_BUCKETS buckets (ok = false: the float pyramid stays the cascade's input).  Bucket 0 lies below and bucket
// nb - 1 above every threshold's bucket: values outside the thresholds' range clamp into records without thresholds.
// (A threshold of exactly 0 is <= every v >= -0: it is counted in every record's `lo` and has no slot.)
// Exact for EVERY finite v (and -0.0 == +0.0) provided no threshold is negative — then a negative v is below all of
// them, rank 0 — which buildRankTables also requires (a model with a negative threshold keeps the float cascade).
constexpr int RANK_WINDOW = 7;
constexpr int RANK_MAX_BUCKETS = 4096;

struct RankChan
{
    int32_t shift, base, nb; // bucket function
    int32_t recOff;          // first record of this channel in RankTables::rec
    int32_t nThr;            // distinct thresholds
    int32_t thrOff;          // first of them in RankTables::thr (host side only: rankOfThreshold)
    int32_t pad_[2];
};

// Record layout (round 3): dword 0 = t[0] | lo << 16, dwords 1..3 = t[1] | t[2] << 16, ...  Low keys are < 0x8000 and an
// unused slot holds RANK_UNUSED = 0x8000, so with X = (low | 0x8000) in both halves of a dword, `X - dword` leaves
// "t <= low" in bits 15 and 31 without a borrow between the halves: seven compares are four subtractions (the device's
// rank_count; full-rate VALU instead of seven v_cmp + v_addc, which issue at half rate on gfx950).
constexpr uint16_t RANK_UNUSED = 0x8000;
struct RankRec // 16 bytes
{
    uint16_t t0;
    uint16_t lo;
    uint16_t tr[RANK_WINDOW - 1];
    uint16_t& t(int j) { return j == 0 ? t0 : tr[j - 1]; }
    uint16_t t(int j) const { return j == 0 ? t0 : tr[j - 1]; }
};

struct RankTables
{
    bool ok = false;
    std::string why; // ok == false: what ruled the rank cells out
    std::vector<RankChan> chan;
    std::vector<RankRec> rec;
    std::vector<float> thr; // sorted distinct thresholds, channel after channel
    int maxRec = 0;         // largest per-channel record count: the kernels' LDS budget

    uint32_t rankOfCell(int chn, float v) const;      // host mirror of the device function (tests, op entry)
    uint32_t rankOfThreshold(int chn, float t) const; // k + 1 of the text above (0: never true)
};

// chnOfNode[q] >= 0: node q tests a feature of that channel (fids[q] / (mW * mH)); < 0: not a feature test (leaf)
void buildRankTables(const float* thrs, const int32_t* chnOfNode, size_t nNodes, int nChns, RankTables& out);

// cv::resize of a packed 8-bit image by (scale, scale) as the apps' Resizer calls it (src/app/acf/acf.cpp:117-148): output size and
// the tap tables of k_resize_u8 (OpenCV's published CV_8U algorithm; parity unpinned).
struct ResizeTables
{
    int rows = 0, cols = 0, drows = 0, dcols = 0;
    int mode = 0, isx = 1, isy = 1; // RZ_LINEAR / RZ_AREA / RZ_AREA_INT (kernels.hip.h)
    std::vector<int32_t> xlin, ylin; // 4 ints per output column {sx, a0, a1, two} / row {r0, r1, b0, b1}
    std::vector<int32_t> xrun, yrun; // 2 ints per output column / row {first tap, count}
    std::vector<int32_t> xtap, ytap; // 2 ints per tap {source index, float bits}
};
void resizeDims(int rows, int cols, double scale, int& drows, int& dcols);
int buildResizeTables(int rows, int cols, double scale, ResizeTables& t);

int colorPlanes(const acf_hip_params& p);
int numChannels(const acf_hip_params& p);

} // namespace acfhip
