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

int colorPlanes(const acf_hip_params& p);
int numChannels(const acf_hip_params& p);

} // namespace acfhip
