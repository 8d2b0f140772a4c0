// kernels.hip.h — hand-written gfx950 kernels for the ACF pyramid + cascade.
//
// Layout everywhere: a plane is float[w][h] with h (image-y) contiguous, so a
// wave's 64 lanes run along image-y and every global access below is a
// contiguous 256-byte segment per wave.  All kernels take a batch of frames
// (blockIdx.z or a flattened frame index) — single-frame parallelism is too
// small to fill 256 CUs on the sequential stages.
//
// Arithmetic contract: every expression is written in the association order of
// the reference's toolbox code (citations per kernel) and the file is compiled
// with -ffp-contract=off, so results are bit-identical to the IEEE
// restatement of that code.  The reference's three approximate-instruction
// sites (_mm_rsqrt_ps/_mm_rcp_ps, toolbox/sse.hpp:185-192) use correctly
// rounded 1/sqrt and 1/x here.
#pragma once

#include "host_plan.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acfhip
{

// address-space-qualified pointers for __builtin_amdgcn_global_load_lds (LDS-DMA)
typedef const __attribute__((address_space(1))) void* gptr_t;  // global
typedef __attribute__((address_space(3))) void* lptr_t;        // LDS

// ------------------------------------------------------------------------
// rgbConvert (toolbox/rgbConvertMex.cpp)
// ------------------------------------------------------------------------
struct LuvConsts
{
    float mr[3], mg[3], mb[3];
    float minu, minv, un, vn, cun, cvn;
};

// rgb2luv_sse body (:129-187) when VEC, else the scalar rgb2luv (:69-83); the
// reference picks VEC iff n % 4 == 0.  lTable: 1064 floats built on the host
// exactly as rgb2luv_setup does (:39-58).
template <bool VEC>
__device__ __forceinline__ void luv_px(float r, float g, float b, const float* __restrict__ lTable, const LuvConsts& k, float& L, float& U, float& V)
{
    if (VEC)
    {
        const float x = (r * k.mr[0] + g * k.mg[0]) + b * k.mb[0];
        const float y = (r * k.mr[1] + g * k.mg[1]) + b * k.mb[1];
        const float z = (r * k.mr[2] + g * k.mg[2]) + b * k.mb[2];
        const float zz = 1.0f / (x + (1e-35f + (15.0f * y + 3.0f * z)));
        const float lf = 1024.0f * y;
        const float u = (52.0f * x) * zz - k.cun;
        const float v = (117.0f * y) * zz - k.cvn;
        L = lTable[(int)lf];
        U = L * u - k.minu;
        V = L * v - k.minv;
    }
    else
    {
        const float x = k.mr[0] * r + k.mg[0] * g + k.mb[0] * b;
        const float y = k.mr[1] * r + k.mg[1] * g + k.mb[1] * b;
        float z = k.mr[2] * r + k.mg[2] * g + k.mb[2] * b;
        L = lTable[(int)(y * 1024)];
        z = 1 / (x + 15 * y + 3 * z + (float)1e-35);
        U = L * (13 * 4 * x * z - 13 * k.un) - k.minu;
        V = L * (13 * 9 * y * z - 13 * k.vn) - k.minv;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_rgb2luv(const float* __restrict__ in, float* __restrict__ out,
    const float* __restrict__ lTable, LuvConsts k, int n, int64_t in_fs, int64_t out_fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const float* I = in + int64_t(blockIdx.z) * in_fs;
    float* J = out + int64_t(blockIdx.z) * out_fs;
    const float r = I[i], g = I[i + n], b = I[i + 2 * int64_t(n)];
    float L, U, V;
    luv_px<VEC>(r, g, b, lTable, k, L, U, V);
    J[i] = L;
    J[i + n] = U;
    J[i + 2 * int64_t(n)] = V;
}

// rgb2gray (:241-252); REPL: the 1-plane input was replicated to 3 planes first
// (chnsPyramid.cpp:234-244), i.e. r == g == b.
template <bool REPL>
__global__ void __launch_bounds__(256) k_rgb2gray(const float* __restrict__ in, float* __restrict__ out, int n,
    int64_t in_fs, int64_t out_fs, float mr, float mg, float mb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const float* I = in + int64_t(blockIdx.z) * in_fs;
    const float r = I[i];
    const float g = REPL ? r : I[i + n];
    const float b = REPL ? r : I[i + 2 * int64_t(n)];
    out[int64_t(blockIdx.z) * out_fs + i] = r * mr + g * mg + b * mb;
}

// Replicate one plane to three (chnsPyramid.cpp:242-243), colorSpace "orig".
// rgb2hsv (toolbox/rgbConvertMex.cpp:194-238), nrm = 1: three planes in, H, S, V out.  IEEE divisions; h * float(1 / 6.0).
__global__ void __launch_bounds__(256) k_rgb2hsv(const float* __restrict__ in, float* __restrict__ out, int n, int64_t in_fs, int64_t out_fs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const float* I = in + int64_t(blockIdx.z) * in_fs;
    float* J = out + int64_t(blockIdx.z) * out_fs;
    const float r = I[i], g = I[n + i], b = I[2 * int64_t(n) + i];
    float h, s, v;
    if (r == g && g == b)
    {
        h = 0.f;
        s = 0.f;
        v = r * 1.0f;
    }
    else
    {
        float maxv, minv;
        if (r >= g && r >= b)
        {
            maxv = r;
            minv = g < b ? g : b;
            h = (g - b) / (maxv - minv) + 6;
            if (h >= 6)
            {
                h -= 6;
            }
        }
        else if (g >= r && g >= b)
        {
            maxv = g;
            minv = r < b ? r : b;
            h = (b - r) / (maxv - minv) + 2;
        }
        else
        {
            maxv = b;
            minv = r < g ? r : g;
            h = (r - g) / (maxv - minv) + 4;
        }
        h *= (float)(1 / 6.0);
        s = 1 - minv / maxv;
        v = maxv * 1.0f;
    }
    J[i] = h;
    J[n + i] = s;
    J[2 * int64_t(n) + i] = v;
}

__global__ void __launch_bounds__(256) k_replicate3(const float* __restrict__ in, float* __restrict__ out, int n, int64_t in_fs, int64_t out_fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const float v = in[int64_t(blockIdx.z) * in_fs + i];
    float* J = out + int64_t(blockIdx.z) * out_fs;
    J[i] = v;
    J[i + n] = v;
    J[i + 2 * int64_t(n)] = v;
}

// uint8_t channel planes -> f32 (exact), for the uint8_t cascade body (acfDetect1.cpp:157-166)
__global__ void __launch_bounds__(256) k_widen_u8(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n)
    {
        out[i] = float(in[i]);
    }
}

// ------------------------------------------------------------------------
// Packed 8-bit upright frames -> transposed planar f32.
//
// Restates the image entry of the detector: cvt8UC3To32FC3 = convertTo(CV_32FC3, 1/255) (ACF.cpp:114-119; OpenCV's
// 8u->32f cvtScale works in f32: float(v) * float(1/255.0)), I.t() (ACF.cpp:137,149) and the MatP plane split
// (MatP.cpp:51-73).  The colour conversion of chnsPyramid.cpp:230-263 is applied in registers when MODE asks for it,
// so the planar f32 RGB image never goes to HBM.
//
// A workgroup moves one 64 x 64 pixel tile: rows are read as dwords (coalesced along image-x) into LDS, then every
// lane takes 4 consecutive image-y of one image-x and writes one float4 per output plane (coalesced along image-y).
// The LDS row pitch is 65 dwords, so the 16 row-quads of a wave hit 16 distinct banks.
// ------------------------------------------------------------------------
enum
{
    IG_PLANAR = 0, // nOut planes, plane c = component c (after the ro/go/bo swizzle)
    IG_LUV_VEC = 1, // rgb2luv_sse body
    IG_LUV = 2,     // scalar rgb2luv
    IG_GRAY = 3     // rgb2gray
};

struct IngestArgs
{
    const uint8_t* in;
    float* out;
    const float* lTable;
    LuvConsts k;
    float mr, mg, mb;
    int H, W;        // upright rows, columns
    int cpp;         // bytes per pixel (1, 3, 4)
    int ro, go, bo;  // byte offsets of r, g, b inside a pixel
    int rowStride;   // bytes between image rows
    int64_t in_fs;   // bytes between frames
    int64_t out_fs;  // floats between output frames
    int nOut;        // IG_PLANAR: 1 or 3 planes
    int vecStore;    // H % 4 == 0: float4 stores
};

constexpr int IG_T = 64;
constexpr int IG_PITCH = 260; // bytes; 65 dwords

template <int MODE, bool ALIGNED>
__global__ void __launch_bounds__(256) k_ingest_u8(IngestArgs a)
{
    __shared__ uint32_t tileW[IG_T * IG_PITCH / 4];
    uint8_t* tile = reinterpret_cast<uint8_t*>(tileW);
    const int x0 = blockIdx.x * IG_T, y0 = blockIdx.y * IG_T;
    const int nx = min(IG_T, a.W - x0), ny = min(IG_T, a.H - y0);
    const uint8_t* src = a.in + int64_t(blockIdx.z) * a.in_fs + int64_t(y0) * a.rowStride + int64_t(x0) * a.cpp;
    const int nb = nx * a.cpp;
    if (ALIGNED)
    {
        // base, row stride and frame stride are multiples of 4 (host-checked) and x0 * cpp is a multiple of 64
        const int nd = (nb + 3) >> 2; // the last dword of a row may run into the next row: still inside the frame
        const int lastOk = (y0 + ny == a.H && blockIdx.z == gridDim.z - 1) ? (nb >> 2) : nd; // ... except at the very end
        for (int i = threadIdx.x; i < ny * 64; i += 256)
        {
            const int yy = i >> 6, j = i & 63;
            if (j < nd)
            {
                const uint8_t* rp = src + int64_t(yy) * a.rowStride;
                uint32_t v;
                if (j < lastOk || yy + 1 < ny)
                {
                    v = reinterpret_cast<const uint32_t*>(rp)[j];
                }
                else
                {
                    v = 0;
                    for (int b = 0; b < nb - 4 * j; b++)
                    {
                        v |= uint32_t(rp[4 * j + b]) << (8 * b);
                    }
                }
                tileW[yy * (IG_PITCH / 4) + j] = v;
            }
        }
    }
    else
    {
        for (int i = threadIdx.x; i < ny * 256; i += 256)
        {
            const int yy = i >> 8, j = i & 255;
            if (j < nb)
            {
                tile[yy * IG_PITCH + j] = src[int64_t(yy) * a.rowStride + j];
            }
        }
    }
    __syncthreads();
    const int yq = (threadIdx.x & 15) * 4;
    const float sc = float(1.0 / 255.0);
    float* outF = a.out + int64_t(blockIdx.z) * a.out_fs;
    const int64_t np = int64_t(a.H) * a.W;
    for (int xx = threadIdx.x >> 4; xx < nx; xx += 16)
    {
        float o[3][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int yy = min(yq + j, IG_T - 1);
            const uint8_t* px = tile + yy * IG_PITCH + xx * a.cpp;
            const float r = float(px[a.ro]) * sc, g = float(px[a.go]) * sc, b = float(px[a.bo]) * sc;
            if (MODE == IG_PLANAR)
            {
                o[0][j] = r;
                o[1][j] = g;
                o[2][j] = b;
            }
            else if (MODE == IG_GRAY)
            {
                o[0][j] = r * a.mr + g * a.mg + b * a.mb; // rgbConvertMex.cpp:241-252
            }
            else
            {
                luv_px<MODE == IG_LUV_VEC>(r, g, b, a.lTable, a.k, o[0][j], o[1][j], o[2][j]);
            }
        }
        const int nPl = (MODE == IG_PLANAR) ? a.nOut : (MODE == IG_GRAY ? 1 : 3);
        const int64_t at = int64_t(x0 + xx) * a.H + y0 + yq;
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
            if (c < nPl)
            {
                float* d = outF + c * np + at;
                if (a.vecStore && yq + 3 < ny)
                {
                    *reinterpret_cast<float4*>(d) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
                }
                else
                {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        if (yq + j < ny)
                        {
                            d[j] = o[c][j];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// convTri1 with the pyramid's in-place aliasing (toolbox/convConst.cpp:445-525;
// chnsCompute.cpp:239, chnsPyramid.cpp:404).
//
// Because source and destination are the same buffer in the reference, the
// x tap of column i reads the OUTPUT of column i-1:
//     T_i[y] = nrm*((O_{i-1}[y] + p*I_i[y]) + I_{i+1}[y])      (O_{-1} := I_0)
//     O_i[y] = (T_i[y-1] + p*T_i[y]) + T_i[y+1]                (edges: (1+p)*T)
// a recursion along image-x with a 3-tap exchange along image-y every step.
// It cannot be tiled along x without changing bits, so one workgroup owns a
// whole plane: threads run along y (R interleaved rows each), the previous
// output column stays in registers (the kernel is out-of-place: it never
// re-reads what it wrote), and the y exchange goes through a double-buffered
// LDS column with one barrier per image column.
// ------------------------------------------------------------------------
// ------------------------------------------------------------------------
// The apps' resize to a minimum object width (src/app/acf/acf.cpp:117-148 `Resizer`, GPUDetectionPipeline.cpp:250-266):
// cv::resize of the packed 8-bit image by scale = winSize.width / minWidth, INTER_AREA when reducing, INTER_LINEAR else.
// OpenCV is not part of the reference tree: the arithmetic is the published algorithm of imgproc/resize.cpp for CV_8U,
// written down in DESIGN.md 6b (and restated on the CPU by the test checker) — PARITY UNPINNED.  One thread per
// output pixel (all channels); the tap tables are built on the host in double precision (host_plan.cpp).
//   RZ_LINEAR    x: {sx, a0, a1, two} per column, y: {r0, r1, b0, b1} per row; 11-bit fixed point:
//                (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2 with h = S[sx] * a0 + S[sx + 1] * a1
//   RZ_AREA      fractional scales: per output column / row a run of {source index, float weight}; per source row
//                buf = sum S * alpha (taps ascending, from 0), then sum = beta * buf (first row), sum += beta * buf; cvRound
//   RZ_AREA_INT  integral scales: (a + b + c + d + 2) >> 2 for 2 x 2, else cvRound(int sum * float(1 / area)); cells that
//                reach past the source: cvRound(float(sum) / count) over the pixels inside
// ------------------------------------------------------------------------
enum
{
    RZ_LINEAR = 0,
    RZ_AREA = 1,
    RZ_AREA_INT = 2
};
struct ResizeArgs
{
    const uint8_t* src;
    uint8_t* dst;
    int32_t rows, cols, cn, stride, drows, dcols;
    int64_t src_fs, dst_fs; // bytes per frame
    int32_t mode, isx, isy;
    const int4* xlin;   // RZ_LINEAR [dcols]
    const int4* ylin;   // RZ_LINEAR [drows]
    const int2* xrun;   // RZ_AREA [dcols] {first tap, count}
    const int2* yrun;   // RZ_AREA [drows]
    const int2* xtap;   // RZ_AREA {source column, float bits}
    const int2* ytap;
};
__device__ __forceinline__ uint8_t rz_sat_u8(float v)
{
    const int i = __float2int_rn(v); // cvRound: round half to even
    return uint8_t(min(max(i, 0), 255));
}
__global__ void __launch_bounds__(256) k_resize_u8(ResizeArgs a)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= a.dcols || dy >= a.drows)
    {
        return;
    }
    const uint8_t* __restrict__ S = a.src + int64_t(blockIdx.z) * a.src_fs;
    uint8_t* __restrict__ D = a.dst + int64_t(blockIdx.z) * a.dst_fs + (int64_t(dy) * a.dcols + dx) * a.cn;
    const int cn = a.cn;
    if (a.mode == RZ_LINEAR)
    {
        const int4 X = a.xlin[dx], Y = a.ylin[dy];
        const uint8_t* S0 = S + int64_t(Y.x) * a.stride + X.x * cn;
        const uint8_t* S1 = S + int64_t(Y.y) * a.stride + X.x * cn;
        const int nx = X.w ? cn : 0; // (beyond xmax the second tap is not read: S[sx] * 2048)
        for (int c = 0; c < cn; c++)
        {
            const int h0 = X.w ? S0[c] * X.y + S0[nx + c] * X.z : S0[c] * 2048;
            const int h1 = X.w ? S1[c] * X.y + S1[nx + c] * X.z : S1[c] * 2048;
            D[c] = uint8_t((((Y.z * (h0 >> 4)) >> 16) + ((Y.w * (h1 >> 4)) >> 16) + 2) >> 2);
        }
        return;
    }
    if (a.mode == RZ_AREA_INT)
    {
        const int sy0 = dy * a.isy, sx0 = dx * a.isx;
        const bool partial = sy0 + a.isy > a.rows || sx0 + a.isx > a.cols;
        const int ny = min(a.isy, a.rows - sy0), nx = min(a.isx, a.cols - sx0);
        const float sc = 1.f / float(a.isx * a.isy);
        for (int c = 0; c < cn; c++)
        {
            int sum = 0;
            for (int y = 0; y < ny; y++)
            {
                for (int x = 0; x < nx; x++)
                {
                    sum += S[int64_t(sy0 + y) * a.stride + (sx0 + x) * cn + c];
                }
            }
            const int count = max(ny, 0) * max(nx, 0);
            if (partial)
            {
                D[c] = count > 0 ? rz_sat_u8(float(sum) / float(count)) : uint8_t(0);
            }
            else
            {
                D[c] = (a.isx == 2 && a.isy == 2) ? uint8_t((sum + 2) >> 2) : rz_sat_u8(float(sum) * sc);
            }
        }
        return;
    }
    const int2 xr = a.xrun[dx], yr = a.yrun[dy];
    for (int c = 0; c < cn; c++)
    {
        float sum = 0.f;
        for (int j = 0; j < yr.y; j++)
        {
            const int2 ty = a.ytap[yr.x + j];
            const uint8_t* Sr = S + int64_t(ty.x) * a.stride + c;
            float buf = 0.f;
            for (int k = 0; k < xr.y; k++)
            {
                const int2 tx = a.xtap[xr.x + k];
                buf = buf + float(Sr[tx.x * cn]) * __int_as_float(tx.y);
            }
            const float t = __int_as_float(ty.y) * buf;
            sum = j == 0 ? t : sum + t;
        }
        D[c] = rz_sat_u8(sum);
    }
}

struct SmoothJob
{
    int32_t h, w, nplanes, out_cs; // out_cs: destination column stride (hP)
    int64_t in_off, out_off;       // float offsets inside a frame's source / destination buffer
    int64_t in_ps, out_ps;         // plane strides
};

// Columns are loaded SM_CH at a time, one whole chunk ahead of the chunk being
// filtered, into two register sets that swap roles (main loop unrolled over two
// chunks: no copies).  The main loop is straight-line code: row and column indices
// are clamped instead of guarded and rows beyond the plane store to a dump slot,
// because vmcnt completes in order and the compiler only keeps the next chunk's loads
// in flight across a column step when it sees no branch between them.
#define SM_CH 8

template <int R, bool ALIASED>
__global__ void __launch_bounds__(1024) k_smooth_tri1(const float* __restrict__ in, float* __restrict__ out,
    const SmoothJob* __restrict__ jobs, int64_t in_fs, int64_t out_fs, float p, int ldsStride, float* __restrict__ dump)
{
    extern __shared__ float lds[]; // 2 * ldsStride floats
    const SmoothJob job = jobs[blockIdx.y];
    if ((int)blockIdx.x >= job.nplanes)
    {
        return;
    }
    const int h = job.h, w = job.w;
    const float* __restrict__ I = in + int64_t(blockIdx.z) * in_fs + job.in_off + int64_t(blockIdx.x) * job.in_ps;
    float* __restrict__ O = out + int64_t(blockIdx.z) * out_fs + job.out_off + int64_t(blockIdx.x) * job.out_ps;
    const int tid = threadIdx.x, nt = blockDim.x;
    const float nrm = 1.0f / ((p + 2) * (p + 2));
    const float p1 = 1 + p;
    int yk[R], ym[R], yp[R]; // this thread's rows (clamped) and their neighbours
    bool ok[R];
#pragma unroll
    for (int k = 0; k < R; k++)
    {
        const int y = tid + k * nt;
        ok[k] = y < h;
        yk[k] = min(y, h - 1);
        ym[k] = max(yk[k] - 1, 0);
        yp[k] = min(yk[k] + 1, h - 1);
    }
    float c0[SM_CH][R], c1[SM_CH][R], prev[R], lastIn[R];
#define SM_LOAD(BUF, I0)                                                          \
    _Pragma("unroll") for (int j = 0; j < SM_CH; j++)                             \
    {                                                                             \
        const float* __restrict__ col = I + int64_t(min((I0) + j, w - 1)) * h;    \
        _Pragma("unroll") for (int k = 0; k < R; k++)                             \
        {                                                                         \
            BUF[j][k] = col[yk[k]];                                               \
        }                                                                         \
    }
    // one column: CUR = column i, NXT = column i+1 (already clamped to w-1 by the loads)
#define SM_COL(I_, CUR, NXT)                                                      \
    {                                                                             \
        const int i_ = (I_);                                                      \
        float* Tb = lds + (i_ & 1) * ldsStride;                                   \
        float T[R];                                                               \
        _Pragma("unroll") for (int k = 0; k < R; k++)                             \
        {                                                                         \
            const float Im = CUR[k];                                              \
            const float Ir = NXT[k]; /* column min(i+1, w-1): Ir = Im at the last column (:508-512) */ \
            const float Il = ALIASED ? ((i_ == 0) ? Im : prev[k]) : ((i_ == 0) ? Im : lastIn[k]);     \
            T[k] = nrm * (Il + p * Im + Ir);                                      \
            lastIn[k] = Im;                                                       \
            Tb[yk[k]] = T[k]; /* rows beyond the plane rewrite row h-1 with its own value */          \
        }                                                                         \
        __syncthreads();                                                          \
        float* __restrict__ oc = O + int64_t(i_) * job.out_cs;                    \
        _Pragma("unroll") for (int k = 0; k < R; k++)                             \
        {                                                                         \
            const float tm = Tb[ym[k]], tp = Tb[yp[k]];                           \
            const float mid = tm + p * T[k] + tp;                                 \
            const float top = p1 * T[k] + tp;                                     \
            const float bot = tm + p1 * T[k];                                     \
            const float o = (yk[k] == 0) ? top : ((yk[k] == h - 1) ? bot : mid);  \
            prev[k] = o;                                                          \
            float* __restrict__ dst = ok[k] ? (oc + yk[k]) : (dump + (tid & 63)); \
            *dst = o;                                                             \
        }                                                                         \
    }
#pragma unroll
    for (int k = 0; k < R; k++)
    {
        prev[k] = lastIn[k] = 0.f;
    }
    SM_LOAD(c0, 0);
    int i = 0;
    // main loop: two full chunks per iteration; needs columns i .. i + 2*SM_CH (the lookahead column is clamped)
    for (; i + 2 * SM_CH <= w; i += 2 * SM_CH)
    {
        SM_LOAD(c1, i + SM_CH);
#pragma unroll
        for (int j = 0; j < SM_CH; j++)
        {
            if (j < SM_CH - 1)
            {
                SM_COL(i + j, c0[j], c0[j + 1]);
            }
            else
            {
                SM_COL(i + j, c0[j], c1[0]);
            }
        }
        SM_LOAD(c0, i + 2 * SM_CH);
#pragma unroll
        for (int j = 0; j < SM_CH; j++)
        {
            if (j < SM_CH - 1)
            {
                SM_COL(i + SM_CH + j, c1[j], c1[j + 1]);
            }
            else
            {
                SM_COL(i + SM_CH + j, c1[j], c0[0]);
            }
        }
    }
    // tail: fewer than 2*SM_CH columns left; c0 holds columns i .. i+SM_CH-1 (clamped)
    if (i < w)
    {
        SM_LOAD(c1, i + SM_CH);
#pragma unroll
        for (int j = 0; j < SM_CH; j++)
        {
            if (i + j < w) // uniform
            {
                if (j < SM_CH - 1)
                {
                    SM_COL(i + j, c0[j], c0[j + 1]);
                }
                else
                {
                    SM_COL(i + j, c0[j], c1[0]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SM_CH; j++)
        {
            if (i + SM_CH + j < w) // uniform
            {
                if (j < SM_CH - 1)
                {
                    SM_COL(i + SM_CH + j, c1[j], c1[j + 1]);
                }
                else
                {
                    SM_COL(i + SM_CH + j, c1[j], c1[j]);
                }
            }
        }
    }
#undef SM_LOAD
#undef SM_COL
}

// ------------------------------------------------------------------------
// Image smoothing with 16 bytes per lane and fused consumers (h % 4 == 0, w % 4 == 0).
// A thread owns 4 consecutive image rows of one plane; the recursion along image-x
// is k_smooth_tri1's.  Of the two y neighbours a row needs, three of four are the
// thread's own registers; the first / last row's are the adjacent LANES' (two wave rotates).
// Across waves the recursion would need one value per column and side — a workgroup barrier
// per column, which is what bounded this kernel (1.45k cycles per column step, 3 TB/s).
// out[x][y] depends on out[x-1][y-1 .. y+1] only, so a wave that also carries SV_K = 2 row
// quads (8 rows) of each neighbouring wave computes its own 60 quads correctly for 8 columns
// without hearing from anybody: the error of a stale halo moves inwards one ROW per column.
// The halo lanes' state (the previous column's four outputs) is refreshed from the owning
// waves once per 8-column chunk: one barrier per chunk instead of eight, 6 % redundant lanes.
// Every value an owner lane stores is computed from the same operands in the same order.
//
// Because a thread's four rows are exactly one shrink-4 cell row and two
// half-resolution row pairs, the consumers of the smoothed image are produced here,
// from registers, instead of re-reading the full-resolution planes:
//   SHRINK  the colour channels of the level: addChn's exact 1/4 resample
//           (chnsCompute.cpp:253-256,346-351; imResampleMex.cpp:210-215,312-317):
//           (((A0+A1)+A2)+A3) along x, then the 4-row sum, * r/4 — k_chns's colour branch;
//   HALF    the next real scale's image when it is an exact half (chnsPyramid.cpp:300-316;
//           imResampleMex.cpp:198-215,284-288): ((Ae[2y]+Ao[2y]) + (Ae[2y+1]+Ao[2y+1])) * rk
//           — k_resample_half;
//   FULL    the full-resolution smoothed plane itself, only where something still reads
//           it (the gradient plane; every plane of a scale later scales are resampled from).
// All three are compile-time per launch, so the column loop has no branch.
// ------------------------------------------------------------------------
struct SmoothVecArgs
{
    const float* in;  // [planes][w][h]
    float* sm;        // FULL: smoothed planes, same layout
    float* half;      // HALF: [planes][w/2][h/2]
    float* chns;      // SHRINK: channel z at chns + z * cells
    int64_t in_fs, in_ps, sm_fs, sm_ps, half_fs, half_ps, chns_fs, cells;
    int32_t h, w, plane0; // plane0: first plane of this launch (blockIdx.x is relative to it)
    float p, rkHalf, rq_y;
    float* dump;      // >= 256 floats nobody reads
    // Column segments (blockIdx.y = segment): see "speculative segments" below.  segW = columns per segment (a multiple of
    // 16; >= w: one segment, the plain recursion), warm = warm-up columns before a segment's first (a multiple of 16).
    int32_t segW, warm, nSeg, nPlanes;
    int32_t segStride;   // segments per plane in the state buffers (>= nSeg: the two launches of a scale may cut their planes differently)
    float* specState; // [frame][plane][segment][h]: a segment's state after its warm-up = its guess of column x0 - 1
    float* trueState; // [frame][plane][segment][h]: the previous segment's output column x0 - 1
    int32_t* redo;       // repair launch (nSeg == 1): [frame][plane] != 0 -> this plane is recomputed as one segment; NULL: every plane
    int32_t skipZ;       // >= 0: this launch leaves plane skipZ out (k_smooth_grad runs it); blockIdx.x counts the others
    // GRAD (k_smooth_grad): gradMag of the smoothed plane from the chain's registers — M and O of column i - 1 leave when
    // column i has been smoothed; the smoothed plane itself is then only written where a later scale is resampled from it
    float* gM;           // [frame] M, plain [w][h] (nybM == 0) or in 64-column x 16-row blocks (k_grad_mag_vec<true>'s layout)
    float* gO;
    const float* acos;   // GM_ACOS_N floats (index 0 of the table = entry 10010)
    int64_t mo_fs;       // frame stride of M / O in floats
    int32_t nybM, full;
    // TRIX (k_smooth_grad_tri): convTri's x pass (r = 5) over M rides on the same chain — U leaves in k_tri_x5v<true>'s blocked layout
    float* tU;           // [frame] U blocks
    int64_t u_fs;        // frame stride of U in floats
    int32_t nybU;        // (h + 8 + 15) / 16
};

__device__ __forceinline__ float wave_rol1(float v)
{
    // (bound_ctrl: every lane of a rotate has a source, so `old` is never used — without it the compiler writes a
    // v_mov_b32 vD, 0 before every rotate and cannot fold the DPP operand into the instruction that consumes it)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, true)); // lane l <- lane l+1 (63 <- 0)
}
__device__ __forceinline__ float wave_ror1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, true)); // lane l <- lane l-1 (0 <- 63)
}

// Keep scalar / vector values materialised at this point: stops the compiler from sinking the loads that produce them
// into data-dependent selects (which turns straight-line select code into branches with a memory wait in every arm).
// Also used to force a wave-uniform value into a VGPR: a VALU instruction with an SGPR operand issues at 1.7x the cost
// of one without on gfx950 (profiles/ubench/valu_rate.hip).
#define ACF_PIN_V(x) asm volatile("" : "+v"(x))

#define SV_CH 8
constexpr int SV_K = 2;              // halo quads per side: 4 * SV_K rows = SV_CH columns of independence
constexpr int SV_OWN = 64 - 2 * SV_K; // quads a wave owns
constexpr int SV_MAXW = 10;          // waves per plane at most: planes of up to 4 * SV_MAXW * SV_OWN = 2400 rows (a 4K frame: 9 waves)
// Speculative segments.  The recursion along image-x is a contraction: column i depends on column i - 1 through
// nrm * (1, 2, 1) = a factor 1/4 (convConst.cpp:445-525 with p = 2), so the influence of whatever a chain STARTED from
// shrinks fourfold per column and is below the last bit of every float after ~15-25 columns; once two chains agree in
// every bit they agree for ever (same inputs, same state, same instructions).  One plane is a chain of w steps with ~5
// waves: at 1080p a launch of 96 frames keeps 1.4 waves per SIMD busy, bound by the latency of a column step.  So the
// plane is cut into segments; segment s starts `warm` columns early from the border formula (Il = Im, what column 0 does),
// discards what it computes there, and from its first own column on emits the same bits as the single chain — PROVIDED
// its state at the end of the warm-up equals the previous segment's last output, which is not assumed but checked:
// both are written to side buffers, k_smooth_verify compares them bit for bit, and a plane with any difference is
// recomputed as one chain by a second launch of this kernel (`redo`) before anything reads it.  Exactness therefore does
// not rest on the contraction argument; only speed does (no repair has been observed with warm >= 32).
#define GM_ACOS_N 20020
// gradMag's two reciprocals (gradientMex.cpp:209-219 with exact arithmetic): m = min(1 / sqrt(m2), 1e10),
// M = 1 / m, every operation rounded as IEEE.  gm_inv_ieee is that text; the compiler's expansion of it is ~36 VALU
// instructions per pixel (a correctly rounded sqrt with range scaling, two divisions with v_div_scale / v_div_fmas /
// v_div_fixup).  gm_inv_fast returns the same two floats for EVERY finite m2 >= 0 — acf_hip_selftest_gradmag compares the
// two over all 2^31 bit patterns on the device (tests/test_gpu_ops.py) — with one v_rsq_f32 and FMA refinements whose
// residuals are exact: sqrt from the rsq estimate y (m2 * y corrected by its residual), 1 / s refined from the same
// estimate, and 1 / m refined from s (m ~ 1 / s, so s is already within 1 ulp of 1 / m): 17 instructions.  Longer forms
// (a Goldschmidt step before the sqrt residual, second Newton steps) were checked the same way and are not needed.  Inputs
// whose m reaches the clamp (s < 1e-10, incl. m2 = 0 where the estimate is inf and the refinement NaN: `t < 1e10f` is
// false) take the constants.
__device__ __forceinline__ void gm_inv_ieee(float m2, float& m, float& M)
{
    float t = 1.0f / sqrtf(m2);
    m = t < 1e10f ? t : 1e10f;
    M = 1.0f / m;
}
__device__ __forceinline__ void gm_inv_fast(float m2, float& m, float& M)
{
    const float y = __builtin_amdgcn_rsqf(m2);
    const float g = m2 * y, hh = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, m2);
    const float s = __builtin_fmaf(d, hh, g); // RN(sqrt(m2))
    const float e = __builtin_fmaf(-s, y, 1.0f);
    float t = __builtin_fmaf(e, y, y);        // RN(1 / s) ...
    // ... but for s = 2^k (1 - 2^-24) (mantissa all ones: m2 just below a power of 4), where 1 / s = 2^-k (1 + 2^-24 + 2^-48 ..)
    // lies a hair above a tie that y (1 + e) can hit exactly and round to even: the answer there is nextup(2^-k), whose bit
    // pattern is 0x7f000000 - bits(s).  (The only inputs the exhaustive comparison found before this line: 196 of 2^31.)
    const uint32_t sb = __float_as_uint(s);
    t = (sb & 0x7fffffu) == 0x7fffffu ? __uint_as_float(0x7f000000u - sb) : t;
    const bool in = t < 1e10f;
    const float e2 = __builtin_fmaf(-t, s, 1.0f);
    const float q = __builtin_fmaf(e2, s, s); // RN(1 / t)
    m = in ? t : 1e10f;
    M = in ? q : 1.0f / 1e10f;
}
// bit patterns first .. first + count - 1 (as m2): mismatches of gm_inv_fast against gm_inv_ieee; bad[0] = their number,
// bad[1] = the smallest mismatching pattern
__global__ void __launch_bounds__(256) k_gm_inv_selftest(uint32_t first, unsigned long long count, unsigned long long* __restrict__ bad)
{
    unsigned long long nb = 0, lo = ~0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256)
    {
        const uint32_t bits = first + uint32_t(i);
        const float x = __uint_as_float(bits);
        float m0, M0, m1, M1;
        gm_inv_ieee(x, m0, M0);
        gm_inv_fast(x, m1, M1);
        if (__float_as_uint(m0) != __float_as_uint(m1) || __float_as_uint(M0) != __float_as_uint(M1))
        {
            nb++;
            lo = lo < bits ? lo : bits;
        }
    }
    if (nb)
    {
        atomicAdd(&bad[0], nb);
        atomicMin(&bad[1], lo);
    }
}
template <bool FULL, bool HALF, bool SHRINK, bool GRAD = false, bool TRIX = false>
__device__ __forceinline__ void smooth_vec_body(const SmoothVecArgs& a, float* lds, int z, const float* acosT = nullptr)
{
    const int h = a.h, w = a.w, nq = h >> 2;
    const int seg = blockIdx.y;
    const int x0 = seg * a.segW, x1 = min(x0 + a.segW, w), xs = max(x0 - a.warm, 0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nWv = blockDim.x >> 6;
    const int qraw = wv * SV_OWN + lane - SV_K;
    const bool valid = lane >= SV_K && lane < 64 - SV_K && qraw < nq; // this lane owns quad qraw; the others are halo / idle
    const int qc = min(max(qraw, 0), nq - 1);
    const int64_t f = blockIdx.z;
    const float* __restrict__ I = a.in + f * a.in_fs + int64_t(z) * a.in_ps + 4 * qc;
    float* __restrict__ Of = FULL ? a.sm + f * a.sm_fs + int64_t(z) * a.sm_ps + 4 * qc : nullptr;
    float* __restrict__ Oh = HALF ? a.half + f * a.half_fs + int64_t(z) * a.half_ps + 2 * qc : nullptr;
    float* __restrict__ Oc = SHRINK ? a.chns + f * a.chns_fs + int64_t(z) * a.cells + qc : nullptr;
    const int hb = h >> 1, hc = h >> 2;
    const float p = a.p, nrm = 1.0f / ((p + 2) * (p + 2)), p1 = 1 + p;
    const bool first = qraw == 0, last = qraw == nq - 1;
    // exchange slots: lds[parity][wave][side][SV_K quads][4]; side 0 = the wave's first owned quads, 1 = its last
    const int ownSide = lane < 32 ? 0 : 1;
    const int ownIdx = lane < 32 ? lane - SV_K : lane - (64 - 2 * SV_K);      // 0 .. SV_K-1 on the edge lanes
    const bool ownEdge = (lane >= SV_K && lane < 2 * SV_K) || (lane >= 64 - 2 * SV_K && lane < 64 - SV_K);
    const bool haloLo = lane < SV_K && wv > 0, haloHi = lane >= 64 - SV_K && wv + 1 < nWv;
    const int srcWave = haloLo ? wv - 1 : wv + 1, srcSide = haloLo ? 1 : 0, srcIdx = haloLo ? lane : lane - (64 - SV_K);
    float prev[4] = { 0.f, 0.f, 0.f, 0.f }, acc[4] = { 0.f, 0.f, 0.f, 0.f };
    float pp[4] = { 0.f, 0.f, 0.f, 0.f }; // GRAD: the smoothed column before `prev`
    float* __restrict__ gMq = nullptr;
    float* __restrict__ gOq = nullptr;
    if (GRAD)
    {
        // the lane's quad in the M / O planes: blocked [x >> 6][y >> 4][x & 63][y & 15] (nybM > 0) or plain [x][y]
        const int64_t qo = a.nybM > 0 ? int64_t(((uint32_t(4 * qc) >> 4) << 10) + (uint32_t(4 * qc) & 15u)) : int64_t(4 * qc);
        gMq = a.gM + f * a.mo_fs + qo;
        gOq = a.gO + f * a.mo_fs + qo;
    }
    // TRIX: convTri's x pass over M (convConst.cpp:347-442 with r = 5, s = 1; k_tri_x5v's arithmetic per row: T += Il + Ir - 2 * Im,
    // U += nrm * T) on the chain that produces M — ONE segment only (running sums have no warm-up).  M's column c enters a ring of
    // sixteen columns (slot c & 15: static, the loop advances 16 columns per iteration and I0 % 16 is the chunk's PH) and at once
    // pays for output column j = c - 5 = {M[c - 12], M[c - 6], M[c]}.  The head: column 0 when M[0..5] are there (c == 5), and the
    // reflected left taps of j = 1 .. 6 (M[6 - j]) are put into the slots those steps read (10 .. 15: written with their own
    // columns only later).  The last six columns (right taps reflected) re-read M from memory behind the chain.
    float4 ring[16];
    float tT[4] = { 0.f, 0.f, 0.f, 0.f }, tUu[4] = { 0.f, 0.f, 0.f, 0.f };
    float* __restrict__ tUq = nullptr;
    const float triN = 1.0f / (6 * 6 * 6 * 6);
    if (TRIX)
    {
#pragma unroll
        for (int m = 0; m < 16; m++)
        {
            ring[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        tUq = a.tU + f * a.u_fs + int64_t(((uint32_t(4 * qc + 8) >> 4) << 10) + (uint32_t(4 * qc + 8) & 15u));
    }
#define SV_TRI_ADDR(col) (tUq + int64_t((((uint32_t(col) >> 6) * uint32_t(a.nybU)) << 10) + ((uint32_t(col) & 63u) << 4)))
#define SV_TRI_STEP(A_, B_, C_, J_, OKJ_)                                                          \
    {                                                                                             \
        const float a4_[4] = { A_.x, A_.y, A_.z, A_.w };                                          \
        const float b4_[4] = { B_.x, B_.y, B_.z, B_.w };                                          \
        const float c4_[4] = { C_.x, C_.y, C_.z, C_.w };                                          \
        _Pragma("unroll") for (int k = 0; k < 4; k++)                                             \
        {                                                                                         \
            tT[k] += a4_[k] + b4_[k] - 2 * c4_[k];                                                \
            tUu[k] += triN * tT[k];                                                               \
        }                                                                                         \
        *reinterpret_cast<float4*>((valid && (OKJ_)) ? SV_TRI_ADDR(J_) : a.dump + 4 * lane) = make_float4(tUu[0], tUu[1], tUu[2], tUu[3]); \
    }
    // M's column x_ (slot SLOT = x_ & 15, compile-time) has just been computed
#define SV_TRI_PUSH(SLOT, x_, mo)                                                                 \
    {                                                                                             \
        constexpr int s_ = (SLOT);                                                                \
        const float4 mv_ = make_float4(mo[0], mo[1], mo[2], mo[3]);                               \
        ring[s_] = mv_;                                                                           \
        /* (slots <= 5 and 15 are also those of the chain's first columns, where there is no output column yet) */ \
        SV_TRI_STEP(ring[(s_ - 12) & 15], mv_, ring[(s_ - 6) & 15], (x_) - 5, (s_ <= 5 || s_ == 15) ? (x_) >= 6 : true) \
        if (s_ == 5 && (x_) == 5)                                                                 \
        {                                                                                         \
            const float4 e0_ = ring[0];                                                           \
            tT[0] = tUu[0] = e0_.x, tT[1] = tUu[1] = e0_.y, tT[2] = tUu[2] = e0_.z, tT[3] = tUu[3] = e0_.w; \
            _Pragma("unroll") for (int m_ = 1; m_ < 6; m_++)                                      \
            {                                                                                     \
                const float e_[4] = { ring[m_].x, ring[m_].y, ring[m_].z, ring[m_].w };           \
                _Pragma("unroll") for (int k = 0; k < 4; k++)                                     \
                {                                                                                 \
                    tT[k] += e_[k];                                                               \
                    tUu[k] += tT[k];                                                              \
                }                                                                                 \
            }                                                                                     \
            _Pragma("unroll") for (int k = 0; k < 4; k++)                                         \
            {                                                                                     \
                tUu[k] = triN * (2 * tUu[k] - tT[k]);                                             \
                tT[k] = 0;                                                                        \
            }                                                                                     \
            *reinterpret_cast<float4*>(valid ? SV_TRI_ADDR(0) : a.dump + 4 * lane) = make_float4(tUu[0], tUu[1], tUu[2], tUu[3]); \
            _Pragma("unroll") for (int m_ = 0; m_ < 6; m_++)                                      \
            {                                                                                     \
                ring[10 + m_] = ring[5 - m_];                                                     \
            }                                                                                     \
        }                                                                                         \
    }
    float4 c0[SV_CH], c1[SV_CH];
    // gradMag of smoothed column X (gradientMex.cpp:17-87,168-251; k_grad_mag_vec's arithmetic per pixel): LFT / CUR / RGT =
    // the lane's quad in columns max(X - 1, 0), X, min(X + 1, w - 1).  The rows above and below the quad are the
    // neighbouring lanes' (halo lanes hold the neighbouring waves' quads: exact for the nearest row at every step, see
    // SV_REFRESH).  OK_: wave-uniform, false = compute but store to the dump slot (no branch in the column loop).
#define SV_GRAD(X, LFT, CUR, RGT, OK_, SLOT)                                                         \
    {                                                                                             \
        const int x_ = (X);                                                                       \
        const float rx = (x_ == 0 || x_ == w - 1) ? 1.f : .5f;                                    \
        const float gup = wave_ror1(CUR[3]), gdn = wave_rol1(CUR[0]);                             \
        float mo[4], oo[4];                                                                       \
        _Pragma("unroll") for (int k = 0; k < 4; k++)                                             \
        {                                                                                         \
            const bool top_ = first && k == 0, bot_ = last && k == 3;                             \
            const float ry = (top_ || bot_) ? 1.f : .5f;                                          \
            const float ga = (k == 0) ? (first ? CUR[0] : gup) : CUR[k > 0 ? k - 1 : 0];          \
            const float gb = (k == 3) ? (last ? CUR[3] : gdn) : CUR[k < 3 ? k + 1 : 3];           \
            const float gx = (RGT[k] - LFT[k]) * rx;                                              \
            const float gy = (gb - ga) * ry;                                                      \
            const float m2 = gx * gx + gy * gy;                                                   \
            float m;                                                                              \
            gm_inv_fast(m2, m, mo[k]);                                                            \
            float g = (gx * m) * 10000.0f;                                                        \
            g = __int_as_float(__float_as_int(g) ^ (__float_as_int(gy) & 0x80000000));            \
            g = g < 10009.0f ? g : 10009.0f;                                                      \
            g = g > -10009.0f ? g : -10009.0f;                                                    \
            float ov = acosT[(int)g];                                                             \
            if (a.full)                                                                           \
            {                                                                                     \
                ov += (gy < 0) * 3.14159265f;                                                     \
            }                                                                                     \
            oo[k] = ov;                                                                           \
        }                                                                                         \
        const int64_t co = a.nybM > 0 ? int64_t((((uint32_t(x_) >> 6) * uint32_t(a.nybM)) << 10) + ((uint32_t(x_) & 63u) << 4)) : int64_t(x_) * h; \
        const bool st_ = valid && (OK_);                                                          \
        *reinterpret_cast<float4*>(st_ ? gMq + co : a.dump + 4 * lane) = make_float4(mo[0], mo[1], mo[2], mo[3]); \
        *reinterpret_cast<float4*>(st_ ? gOq + co : a.dump + 4 * lane) = make_float4(oo[0], oo[1], oo[2], oo[3]); \
        if (TRIX && (SLOT) >= 0)                                                                  \
        {                                                                                         \
            SV_TRI_PUSH((SLOT) & 15, x_, mo)                                                      \
        }                                                                                         \
    }
#define SV_LOAD(BUF, I0)                                                                          \
    _Pragma("unroll") for (int j = 0; j < SV_CH; j++)                                             \
    {                                                                                             \
        BUF[j] = *reinterpret_cast<const float4*>(I + int64_t(min((I0) + j, w - 1)) * h);         \
    }
    // column i = I0 + JJ (JJ compile-time, I0 % 8 == 0): CUR = column i, NXT = column min(i+1, w-1)
#define SV_COL(EMIT, I0, JJ, CUR, NXT, PH)                                                           \
    {                                                                                             \
        const int i_ = (I0) + (JJ);                                                               \
        const float im[4] = { CUR.x, CUR.y, CUR.z, CUR.w };                                       \
        const float ir[4] = { NXT.x, NXT.y, NXT.z, NXT.w };                                       \
        float T[4];                                                                               \
        _Pragma("unroll") for (int k = 0; k < 4; k++)                                             \
        {                                                                                         \
            const float il = (i_ == xs) ? im[k] : prev[k]; /* Il = Im at i == 0 (convConst.cpp:503-507); a later segment's warm-up starts the same way */ \
            T[k] = nrm * (il + p * im[k] + ir[k]);                                                \
        }                                                                                         \
        const float up = wave_ror1(T[3]); /* row 4q-1: the previous lane's last row */            \
        const float dn = wave_rol1(T[0]); /* row 4q+4: the next lane's first row */               \
        float o[4];                                                                               \
        {                                                                                         \
            const float mid0 = up + p * T[0] + T[1], top0 = p1 * T[0] + T[1];                     \
            o[0] = first ? top0 : mid0;                                                           \
            o[1] = T[0] + p * T[1] + T[2];                                                        \
            o[2] = T[1] + p * T[2] + T[3];                                                        \
            const float mid3 = T[2] + p * T[3] + dn, bot3 = T[2] + p1 * T[3];                     \
            o[3] = last ? bot3 : mid3;                                                            \
        }                                                                                         \
        if (FULL && (EMIT))                                                                       \
        {                                                                                         \
            float* dst = valid ? Of + int64_t(i_) * h : a.dump + 4 * lane;                        \
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);                \
        }                                                                                         \
        if (HALF && (EMIT) && ((JJ) & 1))                                                         \
        {                                                                                         \
            float2 hv;                                                                            \
            hv.x = ((prev[0] + o[0]) + (prev[1] + o[1])) * a.rkHalf;                              \
            hv.y = ((prev[2] + o[2]) + (prev[3] + o[3])) * a.rkHalf;                              \
            float* dst = valid ? Oh + int64_t(i_ >> 1) * hb : a.dump + 2 * lane;                  \
            *reinterpret_cast<float2*>(dst) = hv;                                                 \
        }                                                                                         \
        if (SHRINK && (EMIT))                                                                     \
        {                                                                                         \
            _Pragma("unroll") for (int k = 0; k < 4; k++)                                         \
            {                                                                                     \
                acc[k] = (((JJ) & 3) == 0) ? o[k] : acc[k] + o[k];                                \
            }                                                                                     \
            if (((JJ) & 3) == 3)                                                                  \
            {                                                                                     \
                float* dst = valid ? Oc + int64_t(i_ >> 2) * hc : a.dump + lane;                  \
                *dst = (((acc[0] + acc[1]) + acc[2]) + acc[3]) * a.rq_y;                          \
            }                                                                                     \
        }                                                                                         \
        if (GRAD && (EMIT))                                                                       \
        {                                                                                         \
            /* column i - 1: left neighbour pp (for column 0: itself, see below), right neighbour o = column i; the    */ \
            /* segment's first step has no column i - 1 of its own (the previous segment's extra step emits it):      */ \
            /* computed, not stored                                                                                   */ \
            SV_GRAD(i_ - 1, pp, prev, o, i_ > x0, ((JJ) - 1 + (PH)) & 15)                                             \
        }                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 4; k++)                                             \
        {                                                                                         \
            if (GRAD)                                                                             \
            {                                                                                     \
                pp[k] = (i_ == xs) ? o[k] : prev[k]; /* after the chain's first step pp = prev: column 0's left neighbour is itself */ \
            }                                                                                     \
            prev[k] = o[k];                                                                       \
        }                                                                                         \
    }
    // after a chunk: the halo lanes take the state of the quads they shadow from the waves that own them.  Two slot sets
    // alternate: a wave can only overwrite a set two chunks later, i.e. after the barrier that follows every wave's reads.
#define SV_REFRESH(I0)                                                                            \
    if (nWv > 1)                                                                                  \
    {                                                                                             \
        float* xs = lds + (((I0) >> 3) & 1) * (SV_MAXW * 2 * SV_K * 4);                                 \
        if (ownEdge)                                                                              \
        {                                                                                         \
            *reinterpret_cast<float4*>(xs + ((wv * 2 + ownSide) * SV_K + ownIdx) * 4) = make_float4(prev[0], prev[1], prev[2], prev[3]); \
        }                                                                                         \
        __syncthreads();                                                                          \
        if (haloLo || haloHi)                                                                     \
        {                                                                                         \
            const float4 v_ = *reinterpret_cast<const float4*>(xs + ((srcWave * 2 + srcSide) * SV_K + srcIdx) * 4); \
            prev[0] = v_.x, prev[1] = v_.y, prev[2] = v_.z, prev[3] = v_.w;                       \
        }                                                                                         \
    }
    // PH = I0 % 16 (compile-time: TRIX's ring slots; the loops below advance 16 columns per iteration from a multiple of 16)
#define SV_CHUNK(EMIT, I0, A_, B_, PH)                                                            \
    SV_COL(EMIT, I0, 0, A_[0], A_[1], PH) SV_COL(EMIT, I0, 1, A_[1], A_[2], PH) SV_COL(EMIT, I0, 2, A_[2], A_[3], PH) SV_COL(EMIT, I0, 3, A_[3], A_[4], PH) \
    SV_COL(EMIT, I0, 4, A_[4], A_[5], PH) SV_COL(EMIT, I0, 5, A_[5], A_[6], PH) SV_COL(EMIT, I0, 6, A_[6], A_[7], PH) SV_COL(EMIT, I0, 7, A_[7], B_[0], PH) \
    SV_REFRESH(I0)
    const int64_t stateOff = ((f * a.nPlanes + z) * a.segStride) * int64_t(h) + 4 * qc;
    SV_LOAD(c0, xs);
    int i = xs;
    // warm-up of a later segment (x0 - xs is a multiple of 16): same arithmetic, nothing leaves
    for (; i < x0; i += 2 * SV_CH)
    {
        SV_LOAD(c1, i + SV_CH);
        SV_CHUNK(false, i, c0, c1, 0);
        SV_LOAD(c0, i + 2 * SV_CH);
        SV_CHUNK(false, i + SV_CH, c1, c0, 8);
    }
    if (seg > 0 && valid)
    {
        *reinterpret_cast<float4*>(a.specState + stateOff + int64_t(seg) * h) = make_float4(prev[0], prev[1], prev[2], prev[3]);
    }
    for (; i + 2 * SV_CH <= x1; i += 2 * SV_CH)
    {
        SV_LOAD(c1, i + SV_CH);
        SV_CHUNK(true, i, c0, c1, 0);
        SV_LOAD(c0, i + 2 * SV_CH); // clamped to the last column past the end
        SV_CHUNK(true, i + SV_CH, c1, c0, 8);
    }
    // what is left of the last segment: a chunk (w % 16 >= 8) and / or four columns (w % 8 == 4; column w - 1's right
    // neighbour is itself: the loads are clamped to w - 1)
#define SV_TAIL4(I0, A_, PH) SV_COL(true, I0, 0, A_[0], A_[1], PH) SV_COL(true, I0, 1, A_[1], A_[2], PH) SV_COL(true, I0, 2, A_[2], A_[3], PH) SV_COL(true, I0, 3, A_[3], A_[4], PH)
    if (i + SV_CH <= x1)
    {
        SV_LOAD(c1, i + SV_CH);
        SV_CHUNK(true, i, c0, c1, 0);
        i += SV_CH;
        if (i < x1)
        {
            SV_TAIL4(i, c1, 8);
        }
    }
    else if (i < x1)
    {
        SV_TAIL4(i, c0, 0);
    }
#undef SV_TAIL4
    if (seg + 1 < a.nSeg && valid)
    {
        *reinterpret_cast<float4*>(a.trueState + stateOff + int64_t(seg + 1) * h) = make_float4(prev[0], prev[1], prev[2], prev[3]);
    }
    if (GRAD)
    {
        // gradMag of the segment's last column x1 - 1.  The plane's last column has no right neighbour but itself
        // (gradientMex.cpp:31-33); an inner segment smooths one more column, x1 — the exact continuation of its chain: c0
        // holds the input columns x1, x1 + 1 (segW is a multiple of 16: the loop's last prefetch) — for nothing else.
        if (x1 == w)
        {
            SV_GRAD(w - 1, pp, prev, prev, true, -1)
            if (TRIX)
            {
                // output columns w - 6 .. w - 1: M's columns back from memory (each lane re-reads what it wrote itself); the right tap
                // of column j > w - 6 is the reflected M[2w - 6 - j] (convConst.cpp:408-411)
#define SV_TRI_LD(col) (*reinterpret_cast<const float4*>(gMq + (a.nybM > 0 ? int64_t((((uint32_t(col) >> 6) * uint32_t(a.nybM)) << 10) + ((uint32_t(col) & 63u) << 4)) : int64_t(col) * h)))
                for (int j = w - 6; j < w; j++)
                {
                    const float4 ta = SV_TRI_LD(j - 7), tc = SV_TRI_LD(j - 1), tb = SV_TRI_LD(j > w - 6 ? 2 * w - 6 - j : j + 5);
                    SV_TRI_STEP(ta, tb, tc, j, true)
                }
#undef SV_TRI_LD
            }
        }
        else
        {
            const float4 cur = c0[0], nxt = c0[1];
            const float im[4] = { cur.x, cur.y, cur.z, cur.w };
            const float ir[4] = { nxt.x, nxt.y, nxt.z, nxt.w };
            float T[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                T[k] = nrm * (prev[k] + p * im[k] + ir[k]);
            }
            const float up = wave_ror1(T[3]), dn = wave_rol1(T[0]);
            float o[4];
            o[0] = first ? p1 * T[0] + T[1] : up + p * T[0] + T[1];
            o[1] = T[0] + p * T[1] + T[2];
            o[2] = T[1] + p * T[2] + T[3];
            o[3] = last ? T[2] + p1 * T[3] : T[2] + p * T[3] + dn;
            SV_GRAD(x1 - 1, pp, prev, o, true, -1)
        }
    }
#undef SV_GRAD
#undef SV_TRI_PUSH
#undef SV_TRI_STEP
#undef SV_TRI_ADDR
#undef SV_LOAD
#undef SV_COL
#undef SV_REFRESH
#undef SV_CHUNK
}

// One launch per real scale: the planes that must also be written at full resolution (bit z of fullMask: the gradient
// plane, or every plane of a scale that later scales are resampled from) and the ones that are not run side by side as
// workgroups of the same grid instead of as two launches back to back — a plane is a chain of w column steps, so a launch
// lasts as long as one plane whatever the number of planes.  The flag is workgroup-uniform: each specialisation keeps
// its branch-free column loop.
template <bool HALF>
__global__ void __launch_bounds__(64 * SV_MAXW) k_smooth_vec(SmoothVecArgs a, uint32_t fullMask)
{
    extern __shared__ float lds[]; // [2 chunk parities][SV_MAXW waves][2 sides][SV_K quads][4]: the waves' edge state
    int z = a.plane0 + blockIdx.x;
    if (a.skipZ >= 0 && z >= a.skipZ)
    {
        z++; // (that plane is k_smooth_grad's)
    }
    if (a.redo)
    {
        // repair launch (one workgroup per plane and frame): nothing to do when the plane's segments agreed; otherwise the flag is
        // taken down again for the next call — the flags are zero between calls, so no launch has to clear them first
        if (a.redo[int64_t(blockIdx.z) * a.nPlanes + z] == 0)
        {
            return;
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            a.redo[int64_t(blockIdx.z) * a.nPlanes + z] = 0;
        }
    }
    if ((fullMask >> z) & 1u)
    {
        smooth_vec_body<true, HALF, true>(a, lds, z);
    }
    else
    {
        smooth_vec_body<false, HALF, true>(a, lds, z);
    }
}

// The gradient plane's launch: smoothing (+ colour channel, + half-size image) AND gradMag of the smoothed plane from the
// chain's registers.  The smoothed plane then makes no HBM round trip between the two (8.3 MB written and read per 1080p
// frame and scale 0; it is still written at a scale later scales are resampled from).  Its workgroups carry the acos table
// (80 KB of LDS), which is why the other planes stay in k_smooth_vec's launch (a launch has ONE LDS size).
template <bool HALF>
__global__ void __launch_bounds__(64 * SV_MAXW) k_smooth_grad(SmoothVecArgs a, uint32_t fullMask)
{
    extern __shared__ float lds[]; // the waves' edge state (k_smooth_vec), then the acos table
    const int z = a.plane0;
    if (a.redo)
    {
        if (a.redo[int64_t(blockIdx.z) * a.nPlanes + z] == 0)
        {
            return;
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            a.redo[int64_t(blockIdx.z) * a.nPlanes + z] = 0; // (k_smooth_vec)
        }
    }
    float* acosL = lds + 2 * SV_MAXW * 2 * SV_K * 4;
    for (int i = threadIdx.x; i < GM_ACOS_N; i += blockDim.x)
    {
        acosL[i] = a.acos[i];
    }
    __syncthreads();
    if ((fullMask >> z) & 1u)
    {
        smooth_vec_body<true, HALF, true, true>(a, lds, z, acosL + 10010);
    }
    else
    {
        smooth_vec_body<false, HALF, true, true>(a, lds, z, acosL + 10010);
    }
}

// k_smooth_grad with convTri's x pass over M on the same chain (smooth_vec_body's TRIX): M is written once and not read back by a
// separate x pass (8.3 MB per 1080p frame and one launch less per scale).  One segment per plane; up to 8 waves (1920 rows): the
// ring of sixteen M columns costs 64 registers, which a workgroup of ten waves cannot have.
template <bool HALF>
__global__ void __launch_bounds__(512) k_smooth_grad_tri(SmoothVecArgs a, uint32_t fullMask)
{
    extern __shared__ float lds[]; // (k_smooth_grad's)
    const int z = a.plane0;
    float* acosL = lds + 2 * SV_MAXW * 2 * SV_K * 4;
    for (int i = threadIdx.x; i < GM_ACOS_N; i += blockDim.x)
    {
        acosL[i] = a.acos[i];
    }
    __syncthreads();
    if ((fullMask >> z) & 1u)
    {
        smooth_vec_body<true, HALF, true, true, true>(a, lds, z, acosL + 10010);
    }
    else
    {
        smooth_vec_body<false, HALF, true, true, true>(a, lds, z, acosL + 10010);
    }
}

// k_smooth_vec's segments: spec[f][z][s] (segment s's state after its warm-up) against tru[f][z][s] (segment s - 1's last
// output column), s = 1 .. nSeg - 1, bit for bit; any difference marks the plane for the repair launch.
// (segStride: segments per plane in the buffers; plane zG — the gradient plane, k_smooth_grad's — has nSegG segments, the others nSeg)
__global__ void __launch_bounds__(256) k_smooth_verify(const float* __restrict__ spec, const float* __restrict__ tru, int h, int segStride, int nPlanes,
    int32_t* __restrict__ redo, int force, int nSeg, int zG, int nSegG)
{
    const int64_t plane = int64_t(blockIdx.z) * nPlanes + blockIdx.y;
    const int s = 1 + blockIdx.x;
    if (s >= (int(blockIdx.y) == zG ? nSegG : nSeg))
    {
        return;
    }
    const uint32_t* a = reinterpret_cast<const uint32_t*>(spec) + (plane * segStride + s) * int64_t(h);
    const uint32_t* b = reinterpret_cast<const uint32_t*>(tru) + (plane * segStride + s) * int64_t(h);
    bool bad = force != 0;
    for (int y = threadIdx.x; y < h; y += 256)
    {
        bad = bad || (a[y] != b[y]);
    }
    if (bad)
    {
        redo[plane] = 1;
    }
}

// ------------------------------------------------------------------------
// Image-specific lambdas (chnsPyramid.cpp:341-374): the mean of every channel TYPE at two real scales.  The reference
// takes sum(MatP) = the per-plane cv::sum (f32 data, f64 accumulation) added up in plane order (MatP.cpp:97-106).
// cv::sum's own order of additions is OpenCV's SIMD blocking, which is not reproduced (OpenCV is absent from the
// image); the order HERE (the CPU checker of the tests restates it) is: 256 partial sums over the elements
// i = t (mod 256) in increasing i, then the binary tree partial[t] += partial[t + s], s = 128 .. 1.  Against any other
// order of the same f64 additions the result differs by a few units in the last place of a double (relative 1e-16),
// which moves lambda by the same relative amount.
// One workgroup per (plane, selected level, frame); out[frame][sel][plane] (doubles).
// ------------------------------------------------------------------------
struct SumJob
{
    int64_t off;   // float offset of the level's raw channels inside a frame's channel buffer
    int32_t cells; // hC * wC
    int32_t pad_;
};

__global__ void __launch_bounds__(256) k_plane_sums(const float* __restrict__ chns, int64_t chns_fs, SumJob j0, SumJob j1, int nPlanes, double* __restrict__ out)
{
    __shared__ double part[256];
    const int z = blockIdx.x, sel = blockIdx.y;
    const int64_t f = blockIdx.z;
    const SumJob J = sel ? j1 : j0;
    const float* __restrict__ src = chns + f * chns_fs + J.off + int64_t(z) * J.cells;
    double acc = 0.0;
    for (int i = threadIdx.x; i < J.cells; i += 256)
    {
        acc += double(src[i]);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1)
    {
        if (int(threadIdx.x) < s)
        {
            part[threadIdx.x] += part[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        out[(f * 2 + sel) * nPlanes + z] = part[0];
    }
}

// cv::copyMakeBorder(BORDER_REFLECT) of the interior already written by the
// smoothing kernel (chnsPyramid.cpp:410-424): fills only the border cells.
struct PadJob
{
    int32_t hC, wC, hP, wP, py, px, nplanes;
    int32_t pitch; // cells between columns (hP in the float pyramid, hP rounded up to 8 in the rank pyramid)
    int64_t off;   // level offset in the fused pyramid
};

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    while (i < 0 || i >= n)
    {
        i = (i < 0) ? (-i - 1) : (2 * n - 1 - i);
    }
    return i;
}

// One thread per BORDER cell (a thread per cell of the padded level, the interior ones leaving at once, was the largest kernel of
// cfg 4: pad [16 12] on 30 scales per octave, 4.3 ms per 192 VGA frames).  Per plane the items are: for every column x the
// hP - hC rows above and below the interior (item = x * nb + k), then the hC interior rows of the wP - wC columns left and
// right of it.
template <class T> // float: the fused pyramid; uint16_t: its threshold-rank cells (a copied cell keeps its rank)
__global__ void __launch_bounds__(256) k_pad_reflect(T* __restrict__ pyr, const PadJob* __restrict__ jobs, int64_t fs)
{
    const PadJob j = jobs[blockIdx.y];
    const int nb = j.hP - j.hC, nc = j.wP - j.wC;
    const int partA = j.wP * nb, perPlane = partA + nc * j.hC;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= int64_t(perPlane) * j.nplanes)
    {
        return;
    }
    const int c = int(e / perPlane);
    const int item = int(e - int64_t(c) * perPlane);
    int x, y;
    if (item < partA)
    {
        x = item / nb;
        const int k = item - x * nb;
        y = k < j.py ? k : j.hC + k;
    }
    else
    {
        const int it = item - partA;
        const int q = it / j.hC;
        x = q < j.px ? q : j.wC + q;
        y = j.py + (it - q * j.hC);
    }
    const int sx = x - j.px, sy = y - j.py;
    T* P = pyr + int64_t(blockIdx.z) * fs + j.off + int64_t(c) * j.pitch * j.wP;
    const int rx = reflect_idx(sx, j.wC) + j.px, ry = reflect_idx(sy, j.hC) + j.py;
    P[int64_t(x) * j.pitch + y] = P[int64_t(rx) * j.pitch + ry];
}

// ------------------------------------------------------------------------
// gradMag, d == 1 (toolbox/gradientMex.cpp:17-87,168-251).  acosT points at
// the table's centre (index 0).
// ------------------------------------------------------------------------
// A workgroup owns GM_ROWS
// image rows of one frame and walks along image-x in strips of GM_XT columns:
//  - the 20020-entry acos table (80 KB) is copied into LDS once per workgroup.  From
//    global memory the lookup is a 4-byte gather in which every lane pulls its own
//    128-byte line through a 32 KB L1 — on noise-like gradients that moved ~50x the
//    useful bytes and made the lookup, not the image, the kernel's traffic;
//  - the three x taps slide through registers (one row load per column instead of
//    three), every load of a strip is issued before its first use, and border cases
//    are clamped indices + selects (no branch around a load).
#define GM_XT 8
#define GM_ROWS 384
__global__ void __launch_bounds__(GM_ROWS) k_grad_mag_strip(const float* __restrict__ in, float* __restrict__ M, float* __restrict__ O,
    const float* __restrict__ acosBase, int h, int w, int full, int64_t in_fs, int64_t out_fs, int stripsPerBlock)
{
    __shared__ float acosL[GM_ACOS_N];
    for (int i = threadIdx.x; i < GM_ACOS_N; i += GM_ROWS)
    {
        acosL[i] = acosBase[i];
    }
    __syncthreads();
    const float* acosT = acosL + 10010; // index 0 = centre of the table
    const int y = blockIdx.x * GM_ROWS + threadIdx.x;
    const int yc = min(y, h - 1);
    const float* __restrict__ I = in + int64_t(blockIdx.z) * in_fs;
    const int yu = max(yc - 1, 0), yd = min(yc + 1, h - 1);
    const float ry = (yc == 0 || yc == h - 1) ? 1.f : .5f;
    const int nStrips = (w + GM_XT - 1) / GM_XT;
    const int s0 = blockIdx.y * stripsPerBlock, s1 = min(nStrips, s0 + stripsPerBlock);
    for (int s = s0; s < s1; s++)
    {
        const int x0 = s * GM_XT;
        float c[GM_XT + 2], up[GM_XT], dn[GM_XT];
#pragma unroll
        for (int j = 0; j < GM_XT + 2; j++)
        {
            const int x = min(max(x0 + j - 1, 0), w - 1);
            c[j] = I[int64_t(x) * h + yc];
        }
#pragma unroll
        for (int j = 0; j < GM_XT; j++)
        {
            const int x = min(x0 + j, w - 1);
            up[j] = I[int64_t(x) * h + yu];
            dn[j] = I[int64_t(x) * h + yd];
        }
#pragma unroll
        for (int j = 0; j < GM_XT; j++)
        {
            const int x = x0 + j;
            // grad1 :22-53 — one-sided differences with r = 1 at the first / last column, central * .5 inside
            const float rx = (x == 0 || x == w - 1) ? 1.f : .5f;
            const float gx = (c[j + 2] - c[j]) * rx;
            const float gy = (dn[j] - up[j]) * ry;
            const float m2 = gx * gx + gy * gy;
            float m = 1.0f / sqrtf(m2);
            m = m < 1e10f ? m : 1e10f;
            float g = (gx * m) * 10000.0f;
            g = __int_as_float(__float_as_int(g) ^ (__float_as_int(gy) & 0x80000000));
            g = g < 10009.0f ? g : 10009.0f;
            g = g > -10009.0f ? g : -10009.0f;
            float ov = acosT[(int)g];
            if (full)
            {
                ov += (gy < 0) * 3.14159265f;
            }
            if (x < w && y < h)
            {
                const int64_t o = int64_t(blockIdx.z) * out_fs + int64_t(x) * h + y;
                M[o] = 1.0f / m;
                O[o] = ov;
            }
        }
    }
}

// gradMag with 16 bytes per lane (h % 4 == 0) and the acos table in LDS.  A work item is (frame, strip of
// GMV_XT columns, quad of 4 consecutive rows); items are numbered quad-fastest and dealt to a persistent grid
// (2 workgroups per CU, grid-stride), so a wave reads 1 KB contiguous per column and a workgroup amortises its
// one 80 KB table copy over ~60 items per thread.  The x taps slide through registers as float4, the two
// y-neighbour rows outside the thread's own four come from one scalar load each, M / O leave as float4.
// Measured: with the table in global memory the 4-byte lookups (every lane its own 128-byte line through a
// 32 KB L1) were more than half of the kernel.  Same arithmetic as k_grad_mag_strip per pixel.
#define GMV_XT 4
// GMV_BLOCK threads share one copy of the 80 KB table: one workgroup per CU, 16 waves (256 threads = 2 workgroups of 4
// waves per CU left the loads of a wave exposed).
#define GMV_BLOCK 1024
// BL: M and O leave in 64-column x 16-row blocks ([x >> 6][y >> 4][x & 63][y & 15], 4 KB each; nyb = ceil(h / 16), out_fs the
// blocked frame stride): the layout k_tri_x5v<true> and k_triy_chns<.., true> read, see k_tri_x5v.
template <bool BL>
__global__ void __launch_bounds__(GMV_BLOCK) k_grad_mag_vec(const float* __restrict__ in, float* __restrict__ M, float* __restrict__ O,
    const float* __restrict__ acosBase, int h, int w, int full, int64_t in_fs, int64_t out_fs, int nFrames, int nyb)
{
    __shared__ float acosL[GM_ACOS_N];
    for (int i = threadIdx.x; i < GM_ACOS_N; i += GMV_BLOCK)
    {
        acosL[i] = acosBase[i];
    }
    __syncthreads();
    const float* acosT = acosL + 10010; // index 0 = centre of the table
    const int h4 = h >> 2;
    const int nStrips = (w + GMV_XT - 1) / GMV_XT;
    const int64_t perFrame = int64_t(nStrips) * h4;
    const int64_t total = perFrame * nFrames;
    for (int64_t item = int64_t(blockIdx.x) * GMV_BLOCK + threadIdx.x; item < total; item += int64_t(gridDim.x) * GMV_BLOCK)
    {
        const int f = int(item / perFrame);
        const int rem = int(item - int64_t(f) * perFrame);
        const int strip = rem / h4;
        const int q = rem - strip * h4;
        const int x0 = strip * GMV_XT;
        const int y0 = q * 4;
        const float* __restrict__ I = in + int64_t(f) * in_fs;
        const int yu = max(y0 - 1, 0), yd = min(y0 + 4, h - 1);
        float4 c[GMV_XT + 2];
        float up[GMV_XT], dn[GMV_XT];
#pragma unroll
        for (int j = 0; j < GMV_XT + 2; j++)
        {
            const int x = min(max(x0 + j - 1, 0), w - 1);
            c[j] = *reinterpret_cast<const float4*>(I + int64_t(x) * h + y0);
        }
#pragma unroll
        for (int j = 0; j < GMV_XT; j++)
        {
            const int x = min(x0 + j, w - 1);
            up[j] = I[int64_t(x) * h + yu];
            dn[j] = I[int64_t(x) * h + yd];
        }
#pragma unroll
        for (int j = 0; j < GMV_XT; j++)
        {
            const int x = x0 + j;
            const float rx = (x == 0 || x == w - 1) ? 1.f : .5f;
            const float cur[4] = { c[j + 1].x, c[j + 1].y, c[j + 1].z, c[j + 1].w };
            const float lft[4] = { c[j].x, c[j].y, c[j].z, c[j].w };
            const float rgt[4] = { c[j + 2].x, c[j + 2].y, c[j + 2].z, c[j + 2].w };
            float mo[4], oo[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int y = y0 + k;
                // grad1 :22-58 — one-sided differences (factor 1) at the borders, central * .5 inside
                const float ry = (y == 0 || y == h - 1) ? 1.f : .5f;
                const float a = (k == 0) ? ((y == 0) ? cur[0] : up[j]) : cur[k - 1];
                const float b = (k == 3) ? ((y == h - 1) ? cur[3] : dn[j]) : cur[k + 1];
                const float gx = (rgt[k] - lft[k]) * rx;
                const float gy = (b - a) * ry;
                const float m2 = gx * gx + gy * gy;
                float m;
                gm_inv_fast(m2, m, mo[k]); // m = min(1 / sqrt(m2), 1e10), M = 1 / m: the IEEE results, see gm_inv_fast
                float g = (gx * m) * 10000.0f;
                g = __int_as_float(__float_as_int(g) ^ (__float_as_int(gy) & 0x80000000));
                g = g < 10009.0f ? g : 10009.0f;
                g = g > -10009.0f ? g : -10009.0f;
                float ov = acosT[(int)g];
                if (full)
                {
                    ov += (gy < 0) * 3.14159265f;
                }
                oo[k] = ov;
            }
            if (x < w)
            {
                const int64_t o = int64_t(f) * out_fs +
                    (BL ? int64_t((((uint32_t(x) >> 6) * uint32_t(nyb) + (uint32_t(y0) >> 4)) << 10) + ((uint32_t(x) & 63u) << 4) + (uint32_t(y0) & 15u))
                        : int64_t(x) * h + y0);
                *reinterpret_cast<float4*>(M + o) = make_float4(mo[0], mo[1], mo[2], mo[3]);
                *reinterpret_cast<float4*>(O + o) = make_float4(oo[0], oo[1], oo[2], oo[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------
// convTri radius r, x pass (toolbox/convConst.cpp:347-442): second-order
// running sums along image-x, one thread per image row.  Writes U (the
// per-column vector the reference hands to convTriY).
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tri_x(const float* __restrict__ in, float* __restrict__ Uo, int h, int w, int rad, int64_t fs)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h)
    {
        return;
    }
    const float* I = in + int64_t(blockIdx.z) * fs + y;
    float* Uc = Uo + int64_t(blockIdx.z) * fs + y;
    const int r = rad + 1;
    const float nrm = 1.0f / (r * r * r * r);
    float T, U;
    U = T = I[0];
    for (int i = 1; i < r; i++)
    {
        T += I[int64_t(i) * h];
        U += T;
    }
    U = nrm * (2 * U - T);
    T = 0;
    Uc[0] = U;
    int i = 1;
    // head: i <= r  (Il reflected)
    for (; i < w && (i <= r || i > w - r); i++)
    {
        const float Il = (i <= r) ? I[int64_t(r - i) * h] : I[int64_t(i - 1 - r) * h];
        const float Im = I[int64_t(i - 1) * h];
        const float Ir = (i > w - r) ? I[int64_t(2 * w - r - i) * h] : I[int64_t(i - 1 + r) * h];
        T += Il + Ir - 2 * Im;
        U += nrm * T;
        Uc[int64_t(i) * h] = U;
    }
    // body: r < i <= w - r.  Loads do not depend on the recurrence: TX_CH columns (3*TX_CH loads) are
    // requested one whole chunk ahead of the chunk being summed, in two register sets that swap roles
    // (loop unrolled 2x: no copies, so no wait for the set still in flight).
    constexpr int TX_CH = 8;
#define TX_LOAD(A_, B_, C_, I0)                                  \
    _Pragma("unroll") for (int j = 0; j < TX_CH; j++)            \
    {                                                            \
        const int ii = min((I0) + j, w - r); /* clamped: a chunk past the body re-reads valid columns, unused */ \
        A_[j] = I[int64_t(ii - 1 - r) * h];                      \
        B_[j] = I[int64_t(ii - 1 + r) * h];                      \
        C_[j] = I[int64_t(ii - 1) * h];                          \
    }
#define TX_SUM(A_, B_, C_, I0)                                   \
    _Pragma("unroll") for (int j = 0; j < TX_CH; j++)            \
    {                                                            \
        T += A_[j] + B_[j] - 2 * C_[j];                          \
        U += nrm * T;                                            \
        Uc[int64_t((I0) + j) * h] = U;                           \
    }
    if (i + TX_CH - 1 <= w - r)
    {
        float a0[TX_CH], b0[TX_CH], c0[TX_CH], a1[TX_CH], b1[TX_CH], c1[TX_CH];
        TX_LOAD(a0, b0, c0, i);
        for (; i + 2 * TX_CH - 1 <= w - r; i += 2 * TX_CH)
        {
            TX_LOAD(a1, b1, c1, i + TX_CH);
            TX_SUM(a0, b0, c0, i);
            TX_LOAD(a0, b0, c0, i + 2 * TX_CH);
            TX_SUM(a1, b1, c1, i + TX_CH);
        }
        if (i + TX_CH - 1 <= w - r)
        {
            TX_SUM(a0, b0, c0, i);
            i += TX_CH;
        }
    }
#undef TX_LOAD
#undef TX_SUM
    for (; i < w; i++)
    {
        const float Il = (i <= r) ? I[int64_t(r - i) * h] : I[int64_t(i - 1 - r) * h];
        const float Im = I[int64_t(i - 1) * h];
        const float Ir = (i > w - r) ? I[int64_t(2 * w - r - i) * h] : I[int64_t(i - 1 + r) * h];
        T += Il + Ir - 2 * Im;
        U += nrm * T;
        Uc[int64_t(i) * h] = U;
    }
}

// convTri x pass for radius 5, h % 4 == 0, w >= 48: 16 bytes per lane.  A thread owns 4 consecutive image rows
// (four independent running-sum chains) and walks along image-x; columns enter a 16-slot register ring of
// float4 exactly once (step i needs columns i-7, i-1, i+5: with the loop unrolled 16x every ring index is
// static), and the 16 columns of the next iteration are requested one iteration ahead into a second register
// set that swaps roles with the first.  Per chain the arithmetic is k_tri_x's: T += Il + Ir - 2*Im; U += nrm*T.
//
// UT: U leaves in the BLOCKED layout k_triy_chns<.., true> reads: per frame [x >> 6][(y + 8) >> 4][x & 63][(y + 8) & 15] — a
// 64-column x 16-row block is 4 KB contiguous, and the blocks are shifted by 8 rows because the y pass takes rows J+8 .. J+23
// per step.  There a wave's step is then ONE contiguous 4 KB read (lane = column: 64 bytes each) instead of 64-byte halves of
// 128-byte lines taken in two consecutive steps through an L1 that holds a sixth of the CU's working set, and the rows arrive
// in the lanes that own the columns: no transposition through LDS.  Here a column step stores 64-byte pieces 4 KB apart; the
// next column's pieces complete the lines in L2.  ufs: frame stride of U in floats, nyb = (h + 8 + 15) / 16.
// With UT the INPUT is blocked too (unshifted: [x >> 6][y >> 4][x & 63][y & 15], frame stride fs, nybM = ceil(h / 16)), as
// k_grad_mag_vec<true> writes it: k_triy_chns's cells then take a block's M and O in one step as well.
template <bool UT>
__global__ void __launch_bounds__(64) k_tri_x5v(const float* __restrict__ in, float* __restrict__ Uo, int h, int w, int64_t fs, int64_t ufs, int nyb, int nybM)
{
    const int h4 = h >> 2;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= h4)
    {
        return;
    }
    const float* __restrict__ I = UT ? in + int64_t(blockIdx.z) * fs + ((uint32_t(4 * q) >> 4) << 10) + (uint32_t(4 * q) & 15u)
                                     : in + int64_t(blockIdx.z) * fs + 4 * q;
    float* __restrict__ Uc = UT ? Uo + int64_t(blockIdx.z) * ufs + ((uint32_t(4 * q + 8) >> 4) << 10) + (uint32_t(4 * q + 8) & 15u)
                                : Uo + int64_t(blockIdx.z) * fs + 4 * q;
    constexpr int r = 6;
    const float nrm = 1.0f / (r * r * r * r);
#define TXV_LD(col)                                                                                                                 \
    (*reinterpret_cast<const float4*>(UT ? I + (((uint32_t(col) >> 6) * uint32_t(nybM)) << 10) + ((uint32_t(col) & 63u) << 4) : I + int64_t(col) * h))
#define TXV_ST(col, v)                                                                                                              \
    (*reinterpret_cast<float4*>(UT ? Uc + (((uint32_t(col) >> 6) * uint32_t(nyb)) << 10) + ((uint32_t(col) & 63u) << 4) : Uc + int64_t(col) * h) = (v))
    float T[4], U[4];
    {
        const float4 v0 = TXV_LD(0);
        U[0] = T[0] = v0.x, U[1] = T[1] = v0.y, U[2] = T[2] = v0.z, U[3] = T[3] = v0.w;
#pragma unroll
        for (int i = 1; i < r; i++)
        {
            const float4 v = TXV_LD(i);
            const float e[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                T[k] += e[k];
                U[k] += T[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            U[k] = nrm * (2 * U[k] - T[k]);
            T[k] = 0;
        }
        TXV_ST(0, make_float4(U[0], U[1], U[2], U[3]));
    }
#define TXV_STEP(A_, B_, C_, col)                                  \
    {                                                              \
        const float a_[4] = { A_.x, A_.y, A_.z, A_.w };            \
        const float b_[4] = { B_.x, B_.y, B_.z, B_.w };            \
        const float c_[4] = { C_.x, C_.y, C_.z, C_.w };            \
        _Pragma("unroll") for (int k = 0; k < 4; k++)              \
        {                                                          \
            T[k] += a_[k] + b_[k] - 2 * c_[k];                     \
            U[k] += nrm * T[k];                                    \
        }                                                          \
        TXV_ST(col, make_float4(U[0], U[1], U[2], U[3]));          \
    }
    // head: i = 1 .. 15 straight from memory (reflected left taps for i <= r)
#pragma unroll
    for (int i = 1; i < 16; i++)
    {
        const float4 a = (i <= r) ? TXV_LD(r - i) : TXV_LD(i - 1 - r);
        const float4 c = TXV_LD(i - 1);
        const float4 b = TXV_LD(i - 1 + r);
        TXV_STEP(a, b, c, i);
    }
    // ring: slot (column & 15); holds columns J-8 .. J+7 at the top of an iteration
    float4 ring[16];
#pragma unroll
    for (int m = 0; m < 16; m++)
    {
        ring[(8 + m) & 15] = TXV_LD(8 + m);
    }
    int J = 16;
    const int lastFast = w - 24; // J + 23 <= w - 1 and every i <= J + 15 is a body column (i <= w - r)
    float4 nx[16], ny[16];
#define TXV_FETCH(SET, J0)                                         \
    _Pragma("unroll") for (int m = 0; m < 16; m++)                 \
    {                                                              \
        SET[m] = TXV_LD(min((J0) + 8 + m, w - 1));                 \
    }
#define TXV_ITER(SET, J0)                                          \
    _Pragma("unroll") for (int jj = 0; jj < 16; jj++)              \
    {                                                              \
        if (jj >= 3)                                               \
        {                                                          \
            ring[(8 + jj - 3) & 15] = SET[jj - 3]; /* column J+8+m enters before step m+3; its slot's old column was last read at step m-1 */ \
        }                                                          \
        TXV_STEP(ring[(jj - 7) & 15], ring[(jj + 5) & 15], ring[(jj - 1) & 15], (J0) + jj); \
    }                                                              \
    ring[(8 + 13) & 15] = SET[13];                                 \
    ring[(8 + 14) & 15] = SET[14];                                 \
    ring[(8 + 15) & 15] = SET[15];
    if (J <= lastFast)
    {
        TXV_FETCH(nx, J);
        for (; J + 16 <= lastFast; J += 32)
        {
            TXV_FETCH(ny, J + 16);
            TXV_ITER(nx, J);
            TXV_FETCH(nx, J + 32); // clamped: past the body this re-reads valid columns that are not used
            TXV_ITER(ny, J + 16);
        }
        if (J <= lastFast)
        {
            TXV_ITER(nx, J);
            J += 16;
        }
    }
    // tail: remaining columns from memory (reflected right taps for i > w - r)
    for (int i = J; i < w; i++)
    {
        const float4 a = TXV_LD(i - 1 - r);
        const float4 c = TXV_LD(i - 1);
        const float4 b = (i > w - r) ? TXV_LD(2 * w - r - i) : TXV_LD(i - 1 + r);
        TXV_STEP(a, b, c, i);
    }
#undef TXV_LD
#undef TXV_ST
#undef TXV_STEP
#undef TXV_FETCH
#undef TXV_ITER
}

// ------------------------------------------------------------------------
// convTriY (toolbox/convConst.cpp:269-297): second-order running sums down
// each column.  One wave owns 64 columns; 64-row slabs are staged through LDS
// so that global reads and writes stay coalesced along image-y while each lane
// walks its own column.  Lane l reads tile row l: row stride TY_LD is odd, so
// lanes hit distinct banks.
// ------------------------------------------------------------------------
#define TY_CH 64
#define TY_MAXR 16
#define TY_LD (TY_CH + 2 * TY_MAXR + 3)

__global__ void __launch_bounds__(64) k_tri_y(const float* __restrict__ Ui, float* __restrict__ So, int h, int w, int rad, int64_t fs)
{
    __shared__ float tin[64 * TY_LD];
    __shared__ float tout[64 * (TY_CH + 1)];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64;
    const int ncol = min(64, w - x0);
    const float* I = Ui + int64_t(blockIdx.z) * fs + int64_t(x0) * h;
    float* O = So + int64_t(blockIdx.z) * fs + int64_t(x0) * h;
    const int r = rad + 1;
    const int r0 = r - 1, r1 = r + 1, r2 = 2 * h - r, h0 = r + 1, h1 = h - r + 1;
    const int back = r1, ahead = r0; // rows needed behind / ahead of j
    float t = 0, u = 0;
    for (int yb = 0; yb < h; yb += TY_CH)
    {
        // stage rows [lo, hi) of 64 columns
        const int lo = max(0, yb - back), hi = min(h, yb + TY_CH + ahead + 1);
        __syncthreads();
        for (int c = 0; c < ncol; c++)
        {
            for (int yy = lo + lane; yy < hi; yy += 64)
            {
                tin[c * TY_LD + (yy - lo)] = I[int64_t(c) * h + yy];
            }
        }
        __syncthreads();
        if (lane < ncol)
        {
            const float* col = tin + lane * TY_LD - lo; // col[row]
            float* oc = tout + lane * (TY_CH + 1);
            int j = yb;
            const int jend = min(h, yb + TY_CH);
            if (j == 0)
            {
                u = t = col[0];
                for (int q = 1; q < r; q++)
                {
                    t += col[q];
                    u += t;
                }
                u = 2 * u - t;
                t = 0;
                oc[0] = u;
                j = 1;
            }
            for (; j < jend; j++)
            {
                const float a = (j < h0) ? col[r - j] : col[j - r1];
                const float b = (j < h1) ? col[r0 + j] : col[r2 - j];
                t += a + b - 2 * col[j - 1];
                u += t;
                oc[j - yb] = u;
            }
        }
        __syncthreads();
        const int rows = min(TY_CH, h - yb);
        for (int c = 0; c < ncol; c++)
        {
            if (lane < rows)
            {
                O[int64_t(c) * h + yb + lane] = tout[c * (TY_CH + 1) + lane];
            }
        }
    }
}

// convTriY for radius 5 (the normalisation radius every model uses; toolbox/convConst.cpp:269-344: the second-order running
// sums u += t += I[j - r1] + I[r2 - j] - 2 * I[j - 1] down a column, reflected taps at both ends), h % 4 == 0, h >= 48: one lane
// owns one image column and walks down it, its last 16 rows in a register ring; global accesses are staged through LDS.  With
// a lane per column a float4 load straight from memory (round 1's k_tri_y5, deleted in round 5) touches 64 different 128-byte lines, 8 KB of lines per wave; with ~28 waves per CU they do not survive in the L1 between the
// eight loads that use them, so each 16-byte access re-fetched its line from L2 (PMC: 2x the algorithmic HBM traffic,
// L2->L1 traffic ~8x).  Here a wave (64 adjacent columns) moves 16 rows at a time with lanes along image-y — 4 columns
// x 64 contiguous bytes per instruction — through a [64 columns][20 floats] LDS buffer, and each lane then takes its
// own column's 16 rows as four ds_read_b128 (20-float pitch: conflict-free).  Outputs go back the same way.
constexpr int TY_PITCH = 20;
__global__ void __launch_bounds__(256) k_tri_y5s(const float* __restrict__ Ui, float* __restrict__ So, int h, int w, int64_t fs)
{
    __shared__ float ty_lds[4][2][64 * TY_PITCH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 4 + wv) * 64;
    if (x0 >= w)
    {
        return;
    }
    float* inb = ty_lds[wv][0];
    float* outb = ty_lds[wv][1];
    const int x = min(x0 + lane, w - 1); // lanes past the last column duplicate it and never store
    const bool own = x0 + lane < w;
    const float* __restrict__ U0 = Ui + int64_t(blockIdx.z) * fs;
    float* __restrict__ S0 = So + int64_t(blockIdx.z) * fs;
    const float* __restrict__ col = U0 + int64_t(x) * h;
    float* __restrict__ out = S0 + int64_t(x) * h;
    // cooperative mapping: instruction q moves columns 4q + (lane >> 4), rows base + (lane & 15)
    const int cl = lane >> 4, rl = lane & 15;
    constexpr int r = 6, r0 = 5, r1 = 7, h0 = 7;
    const int r2 = 2 * h - r, h1 = h - r + 1;
    float t, u;
    // rows 0..15: the reference's head (reflected taps), straight from memory
    u = t = col[0];
#pragma unroll
    for (int q = 1; q < r; q++)
    {
        t += col[q];
        u += t;
    }
    u = 2 * u - t;
    t = 0;
    float o[16];
    o[0] = u;
#pragma unroll
    for (int j = 1; j < 16; j++)
    {
        const float a = (j < h0) ? col[r - j] : col[j - r1];
        const float b = col[r0 + j];
        t += a + b - 2 * col[j - 1];
        u += t;
        o[j] = u;
    }
    // store rows J..J+15 held in o[] through the LDS buffer
#define TY_STORE(J0)                                                                                        \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            *reinterpret_cast<float4*>(outb + lane * TY_PITCH + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]); \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        _Pragma("unroll") for (int q = 0; q < 16; q++)                                                      \
        {                                                                                                   \
            const int c_ = 4 * q + cl;                                                                      \
            const float v_ = outb[c_ * TY_PITCH + rl];                                                      \
            if (x0 + c_ < w)                                                                                \
            {                                                                                               \
                S0[int64_t(x0 + c_) * h + (J0) + rl] = v_;                                                  \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
    }
    // request rows R0..R0+15 of the wave's 64 columns (16 coalesced loads) into g[]
#define TY_FETCH(G, R0)                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 16; q++)                                                          \
    {                                                                                                       \
        G[q] = U0[int64_t(min(x0 + 4 * q + cl, w - 1)) * h + (R0) + rl];                                    \
    }
    // hand g[] to the owning lanes: N[q] = rows R0+4q .. R0+4q+3 of this lane's column
#define TY_TAKE(G, N)                                                                                       \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; q++)                                                      \
        {                                                                                                   \
            inb[(4 * q + cl) * TY_PITCH + rl] = G[q];                                                       \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            N[q] = *reinterpret_cast<const float4*>(inb + lane * TY_PITCH + 4 * q);                         \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
    }
    TY_STORE(0);
    // ring: slot (row & 15); holds rows J-8 .. J+7 at the top of an iteration
    float ring[16];
    float g[16];
    float4 nx[4];
    {
        TY_FETCH(g, 8);
        TY_TAKE(g, nx); // rows 8..23
        ring[8] = nx[0].x, ring[9] = nx[0].y, ring[10] = nx[0].z, ring[11] = nx[0].w;
        ring[12] = nx[1].x, ring[13] = nx[1].y, ring[14] = nx[1].z, ring[15] = nx[1].w;
        ring[0] = nx[2].x, ring[1] = nx[2].y, ring[2] = nx[2].z, ring[3] = nx[2].w;
        ring[4] = nx[3].x, ring[5] = nx[3].y, ring[6] = nx[3].z, ring[7] = nx[3].w;
    }
    int J = 16;
    const int lastFast = h - 24; // J + 23 <= h - 1 and every j <= J + 15 < h1
    if (J <= lastFast)
    {
        TY_FETCH(g, J + 8);
        TY_TAKE(g, nx); // rows J+8 .. J+23
    }
    for (; J <= lastFast; J += 16)
    {
        const bool more = J + 16 <= lastFast;
        if (more)
        {
            TY_FETCH(g, J + 24); // next iteration's rows, in flight during this iteration's recurrence
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++)
            {
                const int jj = 4 * q + s2; // j = J + jj, J % 16 == 0
                if (s2 == 3)
                {
                    // rows J+8+4q .. J+11+4q replace rows J-8+4q .. J-5+4q (last used as `a` one step ago)
                    ring[(8 + 4 * q) & 15] = nx[q].x;
                    ring[(9 + 4 * q) & 15] = nx[q].y;
                    ring[(10 + 4 * q) & 15] = nx[q].z;
                    ring[(11 + 4 * q) & 15] = nx[q].w;
                }
                const float a = ring[(jj - 7) & 15];
                const float b = ring[(jj + 5) & 15];
                const float cc = ring[(jj - 1) & 15];
                t += a + b - 2 * cc;
                u += t;
                o[jj] = u;
            }
        }
        TY_STORE(J);
        if (more)
        {
            TY_TAKE(g, nx);
        }
    }
#undef TY_STORE
#undef TY_FETCH
#undef TY_TAKE
    // remaining rows (the reflected tail), from memory
    if (own)
    {
        for (int j = J; j < h; j++)
        {
            const float a = col[j - r1];
            const float b = (j < h1) ? col[r0 + j] : col[r2 - j];
            t += a + b - 2 * col[j - 1];
            u += t;
            out[j] = u;
        }
    }
}

// ------------------------------------------------------------------------
// gradMagNorm + gradHist + addChn's exact 1/shrink resample, fused
// (toolbox/gradientMex.cpp:254-275, 278-372, 451-509; chnsCompute.cpp:253-256,
// 303-307, 346-351; toolbox/imResampleMex.cpp:210-215, 312-317).
// One thread per shrink x shrink cell: the 16 pixels of a cell are read once
// (one 16-byte load per column) and every channel of the cell is produced.
// The histogram accumulates in the reference's order: x outer, y inner, O0
// contribution then O1; orientation bins are selected with compares so the six
// accumulators stay in registers.
// ------------------------------------------------------------------------
// gradHist's two bin updates of one pixel (gradientMex.cpp:451-509: H[o0] += m0, H[o1] += m1 with o1 = o0 + 1 wrapped at nO)
// for bins held in registers.  A select per bin and addend — v_cmp, (two wait states,) v_cndmask, v_add: the cost of 3.6 + 1 plain
// instructions on gfx950 (profiles/ubench/valu_rate.hip: `cmp_cnd`) — was more than half of the y pass kernel's issue time.  Here the bin
// takes H + (mask & m) for every b: the mask is all ones for the pixel's bin and 0 elsewhere (a sign-extended bit of 1 << o0:
// v_bfe_i32), so the chosen bin gets the reference's addition and every other bin gets + 0.0f, which changes no bit of a bin —
// they start at +0.0f and only ever add values >= +0 (m0 = m - od * m with 0 <= od < 1, m1 = od * m), so no bin is ever -0.0f.
// o1's masks are o0's moved up by one bin; bin 0 takes bit nO - 1.  (hardBin: m1 = +0.0f, the same argument.)
template <int MAXO>
__device__ __forceinline__ void hist_add2(float (&H)[MAXO], int o0, int nO, float m0, float m1)
{
    const int A = 1 << o0;
    int mk[MAXO];
#pragma unroll
    for (int b = 0; b < MAXO; b++)
    {
        mk[b] = __builtin_amdgcn_sbfe(A, b, 1); // bit b of A, sign-extended: -1 or 0
    }
    const int mkW = __builtin_amdgcn_sbfe(A, nO - 1, 1); // o0 == nO - 1: o1 wraps to bin 0
    const int b0 = __float_as_int(m0), b1 = __float_as_int(m1);
#pragma unroll
    for (int b = 0; b < MAXO; b++)
    {
        const float a0 = __int_as_float(mk[b] & b0);
        const float a1 = __int_as_float((b == 0 ? mkW : mk[b - 1]) & b1);
        H[b] = (H[b] + a0) + a1;
    }
}

struct ChnsArgs
{
    const float* sm;   // smoothed colour planes [d][w][h]
    const float* M;    // gradient magnitude (unnormalised)
    const float* S;    // convTri(M, normRad); unused if !doNorm
    const float* O;
    float* Mn;         // optional tap: normalised magnitude, full resolution (may be null)
    float* chns;       // destination: level's raw channel planes [nC][wC][hC]
    int64_t sm_fs, m_fs, chns_fs;
    int32_t h, w, d;
    int32_t colorEnabled, magEnabled, histEnabled, nOrients, doNorm, full;
    int32_t hardBin;   // softBin < 0: the nearest orientation bin takes the whole magnitude (gradQuantize's interpolate == false, gradientMex.cpp:316-327,355-370)
    int32_t colorDone; // the colour channels were already written by k_smooth_vec: skip them, keep their slots
    float normConst, rq; // rq = (1/S)/(1+1e-6) then /S in the y pass (imResampleMex.cpp:145-157,316)
    float rq_y;
    int32_t nybM;      // blocked M / O (k_triy_chns<.., true>): 16-row blocks per column block, ceil(h / 16); m_fs is then the blocked frame stride
};

template <int S>
__device__ __forceinline__ void chns_load_vec(const float* __restrict__ p, float (&d)[S])
{
    if (S == 4)
    {
        const float4 v = *reinterpret_cast<const float4*>(p);
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
    else if (S == 2)
    {
        const float2 v = *reinterpret_cast<const float2*>(p);
        d[0] = v.x, d[1] = v.y;
    }
    else
    {
#pragma unroll
        for (int i = 0; i < S; i++)
        {
            d[i] = p[i];
        }
    }
}

// ------------------------------------------------------------------------
// convTriY (r = 5) + gradMagNorm + the magnitude channel + gradHist in one kernel: k_tri_y5s's column recurrence, whose
// 16-row x 64-column blocks of S already pass through LDS (lane = column -> coalesced rows), followed at once by k_chns's
// cell arithmetic on that block (lane = one 4 x 4 cell: 64 cells per block) — S never reaches HBM (16.6 MB per 1080p frame
// written and read back by the two-kernel form).  M and O of the block are requested before the recurrence and consumed
// after it.  Values, association order and the histogram's accumulation order are k_tri_y5s's and k_chns's
// (convConst.cpp:347-442; gradientMex.cpp:254-275, 278-372; imResampleMex.cpp:210-215, 312-317).
// Needs shrink 4, h % 4 == 0, h >= 48, normalisation on, the colour channels already written (or disabled), no Mnorm tap.
// ------------------------------------------------------------------------
// UT: U comes in k_tri_x5v<true>'s blocked layout (ufs, nyb as there): a step's rows are four 16-byte loads per lane from one
// contiguous 4 KB block, already in the lane that owns the column.
template <int MAXO, bool UT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_triy_chns(const float* __restrict__ Ui, ChnsArgs ca, int64_t ufs, int nyb)
{
    __shared__ float ty_lds[4][UT ? 1 : 2][64 * TY_PITCH];
    const int h = ca.h, w = ca.w;
    const int64_t fs = ca.m_fs;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 4 + wv) * 64;
    if (x0 >= w)
    {
        return;
    }
    float* inb = ty_lds[wv][0];
    float* outb = ty_lds[wv][UT ? 0 : 1];
    const int x = min(x0 + lane, w - 1); // lanes past the last column duplicate it and never store
    const float* __restrict__ U0 = Ui + int64_t(blockIdx.z) * (UT ? ufs : fs);
    // row j of this lane's column
    const float* __restrict__ colP = U0 + int64_t(x) * h;                                                                    // plain
    const float* __restrict__ colT = U0 + (((uint32_t(x) >> 6) * uint32_t(nyb)) << 10) + ((uint32_t(x) & 63u) << 4);         // blocked
#define TY_U(j) (UT ? colT[((uint32_t((j) + 8) >> 4) << 10) + (uint32_t((j) + 8) & 15u)] : colP[(j)])
    // cooperative mapping: instruction q moves columns 4q + (lane >> 4), rows base + (lane & 15)
    const int cl = lane >> 4, rl = lane & 15;
    constexpr int r = 6, r0 = 5, r1 = 7, h0 = 7;
    const int r2 = 2 * h - r, h1 = h - r + 1;
    float t, u;
    // rows 0..15: the reference's head (reflected taps), straight from memory
    u = t = TY_U(0);
#pragma unroll
    for (int q = 1; q < r; q++)
    {
        t += TY_U(q);
        u += t;
    }
    u = 2 * u - t;
    t = 0;
    float o[16];
    o[0] = u;
#pragma unroll
    for (int j = 1; j < 16; j++)
    {
        const float a = (j < h0) ? TY_U(r - j) : TY_U(j - r1);
        const float b = TY_U(r0 + j);
        t += a + b - 2 * TY_U(j - 1);
        u += t;
        o[j] = u;
    }
    // rows J0 .. J0+15 of S are in o[]: hand them to the cells through the LDS buffer (lane = column -> lane = cell) and
    // finish the cells: gradMagNorm, the magnitude channel and the orientation histogram (k_chns's arithmetic and order).
    const int xcL = lane >> 2, ycL = lane & 3;             // cell of this lane inside the wave's 64 x 16 block
    const int hc = h >> 2;
    const int64_t cellsN = int64_t(hc) * (w >> 2);
    const float* __restrict__ Mf = ca.M + int64_t(blockIdx.z) * fs;
    const float* __restrict__ Of = ca.O + int64_t(blockIdx.z) * fs;
    float* __restrict__ chn = ca.chns + int64_t(blockIdx.z) * ca.chns_fs;
    const int chMag = ca.colorEnabled ? ca.d : 0;            // the colour channels were written by k_smooth_vec
    const float oMult = (float)ca.nOrients / (ca.full ? 2 * 3.14159265f : 3.14159265f);
    const float sInv2 = 1 / (float)4 / (float)4;
    const int nO = ca.nOrients;
    // M and O cells of block k+1 are requested while block k is worked on (requested at the top of the step that consumes
    // them, every step paid a full memory round trip).  M: one register set, re-requested as soon as the normalised
    // magnitudes of the current block exist; O: two sets that swap roles every step (it is live until the histogram).
    float4 mq[4], oqA[4], oqB[4];
    // float offset of (column X, row Y) in a frame of M / O: plain [w][h], or 64 x 16 blocks (UT)
#define TY_MO_OFF(X, Y)                                                                                                       \
    (UT ? (((uint32_t(X) >> 6) * uint32_t(ca.nybM) + (uint32_t(Y) >> 4)) << 10) + ((uint32_t(X) & 63u) << 4) + (uint32_t(Y) & 15u) \
        : uint32_t(X) * uint32_t(h) + uint32_t(Y))
#define TY_M_FETCH(J0)                                                                                      \
    {                                                                                                       \
        const int yq_ = min((J0) + 4 * ycL, h - 4);                                                         \
        _Pragma("unroll") for (int xx = 0; xx < 4; xx++)                                                    \
        {                                                                                                   \
            mq[xx] = *reinterpret_cast<const float4*>(Mf + TY_MO_OFF(min(x0 + 4 * xcL + xx, w - 1), yq_));  \
        }                                                                                                   \
    }
#define TY_O_FETCH(oq, J0)                                                                                  \
    {                                                                                                       \
        const int yq_ = min((J0) + 4 * ycL, h - 4);                                                         \
        _Pragma("unroll") for (int xx = 0; xx < 4; xx++)                                                    \
        {                                                                                                   \
            oq[xx] = *reinterpret_cast<const float4*>(Of + TY_MO_OFF(min(x0 + 4 * xcL + xx, w - 1), yq_));  \
        }                                                                                                   \
    }
#define TY_CELLS(oq, J0, NROWS, JN)                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            *reinterpret_cast<float4*>(outb + lane * TY_PITCH + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]); \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        float sq[4][4], mn[4][4], ov[4][4];                                                                 \
        _Pragma("unroll") for (int xx = 0; xx < 4; xx++)                                                    \
        {                                                                                                   \
            const float4 sv_ = *reinterpret_cast<const float4*>(outb + (4 * xcL + xx) * TY_PITCH + 4 * ycL); \
            sq[xx][0] = sv_.x, sq[xx][1] = sv_.y, sq[xx][2] = sv_.z, sq[xx][3] = sv_.w;                      \
            const float mr_[4] = { mq[xx].x, mq[xx].y, mq[xx].z, mq[xx].w };                                \
            ov[xx][0] = oq[xx].x, ov[xx][1] = oq[xx].y, ov[xx][2] = oq[xx].z, ov[xx][3] = oq[xx].w;          \
            _Pragma("unroll") for (int yy = 0; yy < 4; yy++)                                                \
            {                                                                                               \
                mn[xx][yy] = mr_[yy] * (1.0f / (sq[xx][yy] + ca.normConst)); /* gradMagNorm: M * rcp(S + norm) */ \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        TY_M_FETCH(JN); /* the next block's magnitudes (clamped rows: harmless past the end) */             \
        if (4 * ycL < (NROWS) && x0 + 4 * xcL < w)                                                          \
        {                                                                                                   \
            float* outc = chn + (uint32_t((x0 >> 2) + xcL) * uint32_t(hc) + uint32_t(((J0) >> 2) + ycL));                      \
            if (ca.magEnabled)                                                                               \
            {                                                                                               \
                float C_[4];                                                                                \
                _Pragma("unroll") for (int yy = 0; yy < 4; yy++)                                            \
                {                                                                                           \
                    C_[yy] = ((mn[0][yy] + mn[1][yy]) + mn[2][yy]) + mn[3][yy];                             \
                }                                                                                           \
                outc[int64_t(chMag) * cellsN] = (((C_[0] + C_[1]) + C_[2]) + C_[3]) * ca.rq_y;               \
            }                                                                                               \
            if (ca.histEnabled)                                                                              \
            {                                                                                               \
                float H_[MAXO];                                                                             \
                _Pragma("unroll") for (int b = 0; b < MAXO; b++)                                            \
                {                                                                                           \
                    H_[b] = 0.f;                                                                            \
                }                                                                                           \
                _Pragma("unroll") for (int xx = 0; xx < 4; xx++)                                            \
                {                                                                                           \
                    _Pragma("unroll") for (int yy = 0; yy < 4; yy++)                                        \
                    {                                                                                       \
                        const float ob_ = ov[xx][yy] * oMult;                                               \
                        /* hardBin: o0 = (int)(o + .5f), M0 = m, M1 = 0 — adding that +0.0f to a bin changes no bit */ \
                        int o0_ = ca.hardBin ? (int)(ob_ + .5f) : (int)ob_;                                 \
                        const float od_ = ca.hardBin ? 0.f : ob_ - (float)o0_;                              \
                        o0_ = (o0_ >= nO) ? 0 : o0_;                                                        \
                        int o1_ = o0_ + 1;                                                                  \
                        o1_ = (o1_ == nO) ? 0 : o1_;                                                        \
                        const float m_ = mn[xx][yy] * sInv2;                                                \
                        const float m1_ = od_ * m_;                                                         \
                        const float m0_ = m_ - m1_;                                                         \
                        hist_add2<MAXO>(H_, o0_, nO, m0_, m1_);                                             \
                    }                                                                                       \
                }                                                                                           \
                const int chH_ = chMag + (ca.magEnabled ? 1 : 0);                                            \
                _Pragma("unroll") for (int b = 0; b < MAXO; b++)                                            \
                {                                                                                           \
                    if (b < nO)                                                                             \
                    {                                                                                       \
                        outc[int64_t(chH_ + b) * cellsN] = H_[b];                                           \
                    }                                                                                       \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }
    // request rows R0..R0+15 of the wave's 64 columns: plain layout — 16 coalesced loads into G[], handed to the owning lanes
    // through LDS by TY_TAKE; blocked layout (R0 = J + 8: exactly one block row) — four 16-byte loads per lane, already home
#define TY_FETCH(G, G4, R0)                                                                                 \
    if (UT)                                                                                                 \
    {                                                                                                       \
        const float* p_ = colT + ((uint32_t((R0) + 8) >> 4) << 10);                                         \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            G4[q] = *reinterpret_cast<const float4*>(p_ + 4 * q);                                           \
        }                                                                                                   \
    }                                                                                                       \
    else                                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; q++)                                                      \
        {                                                                                                   \
            G[q] = U0[uint32_t(min(x0 + 4 * q + cl, w - 1)) * uint32_t(h) + uint32_t((R0) + rl)];           \
        }                                                                                                   \
    }
    // N[q] = rows R0+4q .. R0+4q+3 of this lane's column
#define TY_TAKE(G, G4, N)                                                                                   \
    if (UT)                                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            N[q] = G4[q];                                                                                   \
        }                                                                                                   \
    }                                                                                                       \
    else                                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; q++)                                                      \
        {                                                                                                   \
            inb[(4 * q + cl) * TY_PITCH + rl] = G[q];                                                       \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            N[q] = *reinterpret_cast<const float4*>(inb + lane * TY_PITCH + 4 * q);                         \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
    }
    TY_M_FETCH(0);
    TY_O_FETCH(oqA, 0);
    TY_O_FETCH(oqB, 16);
    TY_CELLS(oqA, 0, 16, 16);
    // ring: slot (row & 15); holds rows J-8 .. J+7 at the top of an iteration
    float ring[16];
    float g[16];
    float4 g4[4];
    float4 nx[4];
    {
        TY_FETCH(g, g4, 8);
        TY_TAKE(g, g4, nx); // rows 8..23
        ring[8] = nx[0].x, ring[9] = nx[0].y, ring[10] = nx[0].z, ring[11] = nx[0].w;
        ring[12] = nx[1].x, ring[13] = nx[1].y, ring[14] = nx[1].z, ring[15] = nx[1].w;
        ring[0] = nx[2].x, ring[1] = nx[2].y, ring[2] = nx[2].z, ring[3] = nx[2].w;
        ring[4] = nx[3].x, ring[5] = nx[3].y, ring[6] = nx[3].z, ring[7] = nx[3].w;
    }
    int J = 16;
    const int lastFast = h - 24; // J + 23 <= h - 1 and every j <= J + 15 < h1
    if (J <= lastFast)
    {
        TY_FETCH(g, g4, J + 8);
        TY_TAKE(g, g4, nx); // rows J+8 .. J+23
    }
#define TY_ITER(CO, NO)                                                                             \
    {                                                                                                       \
        const bool more = J + 16 <= lastFast;                                                               \
        TY_O_FETCH(NO, J + 16); /* the NEXT block's orientations */                                         \
        if (more)                                                                                           \
        {                                                                                                   \
            TY_FETCH(g, g4, J + 24); /* next iteration's rows, in flight during this iteration's recurrence */  \
        }                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                       \
        {                                                                                                   \
            _Pragma("unroll") for (int s2 = 0; s2 < 4; s2++)                                                \
            {                                                                                               \
                const int jj = 4 * q + s2; /* j = J + jj, J % 16 == 0 */                                    \
                if (s2 == 3)                                                                                \
                {                                                                                           \
                    /* rows J+8+4q .. J+11+4q replace rows J-8+4q .. J-5+4q (last used as `a` one step ago) */ \
                    ring[(8 + 4 * q) & 15] = nx[q].x;                                                       \
                    ring[(9 + 4 * q) & 15] = nx[q].y;                                                       \
                    ring[(10 + 4 * q) & 15] = nx[q].z;                                                      \
                    ring[(11 + 4 * q) & 15] = nx[q].w;                                                      \
                }                                                                                           \
                const float a_ = ring[(jj - 7) & 15];                                                       \
                const float b_ = ring[(jj + 5) & 15];                                                       \
                const float cc_ = ring[(jj - 1) & 15];                                                      \
                t += a_ + b_ - 2 * cc_;                                                                     \
                u += t;                                                                                     \
                o[jj] = u;                                                                                  \
            }                                                                                               \
        }                                                                                                   \
        TY_CELLS(CO, J, 16, J + 16);                                                                         \
        if (more)                                                                                           \
        {                                                                                                   \
            TY_TAKE(g, g4, nx);                                                                                 \
        }                                                                                                   \
        J += 16;                                                                                            \
    }
    bool inA = false; // which set holds the block at J (block 16 is in B)
    while (J <= lastFast)
    {
        TY_ITER(oqB, oqA);
        inA = true;
        if (J > lastFast)
        {
            break;
        }
        TY_ITER(oqA, oqB);
        inA = false;
    }
#undef TY_ITER
    if (!inA) // wave-uniform; once
    {
#pragma unroll
        for (int xx = 0; xx < 4; xx++)
        {
            oqA[xx] = oqB[xx];
        }
    }
#undef TY_FETCH
#undef TY_TAKE
    // remaining rows (the reflected tail, 8 .. 23 of them), from memory, 16 at a time through the same cell step; the first
    // group's M and O are already in set A
    for (bool firstTail = true; J < h; J += 16, firstTail = false)
    {
        const int nr = min(16, h - J); // a multiple of 4
        if (!firstTail)
        {
            TY_O_FETCH(oqA, J); // (M was requested by the previous group's cell step)
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++)
        {
            const int j = J + jj;
            if (jj < nr) // wave-uniform
            {
                const float a_ = TY_U(j - r1);
                const float b_ = (j < h1) ? TY_U(r0 + j) : TY_U(r2 - j);
                t += a_ + b_ - 2 * TY_U(j - 1);
                u += t;
                o[jj] = u;
            }
        }
        TY_CELLS(oqA, J, nr, J + 16);
    }
#undef TY_U
#undef TY_MO_OFF
#undef TY_M_FETCH
#undef TY_O_FETCH
#undef TY_CELLS
}

// MAXO: compile-time bound of nOrients (6 for every shipped model): the bin update is a select chain over MAXO registers,
// 2 * MAXO * 3 VALU per pixel — half of the kernel's instructions at MAXO = 12.
template <int S, int MAXO>
__global__ void __launch_bounds__(256) k_chns(ChnsArgs a)
{
    const int hc = a.h / S, wc = a.w / S;
    // cells are numbered column-major (yc fastest) and dealt to threads linearly, so every wave is full
    // and its 64 cells are (mostly) one contiguous run of a cell column: 16-byte loads per lane, 1 KB per wave
    const int64_t cells = int64_t(hc) * wc;
    const int64_t cell = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (cell >= cells)
    {
        return;
    }
    const int xc = int(cell / hc);
    const int yc = int(cell - int64_t(xc) * hc);
    const int64_t f = blockIdx.z;
    float* out = a.chns + f * a.chns_fs + int64_t(xc) * hc + yc;
    const int64_t pbase = int64_t(xc * S) * a.h + yc * S;
    int ch = 0;
    if (a.colorEnabled && a.colorDone)
    {
        ch = a.d;
    }
    else if (a.colorEnabled)
    {
        for (int z = 0; z < a.d; z++)
        {
            const float* P = a.sm + f * a.sm_fs + int64_t(z) * a.h * a.w + pbase;
            float C[S];
#pragma unroll
            for (int yy = 0; yy < S; yy++)
            {
                float s = P[yy];
#pragma unroll
                for (int xx = 1; xx < S; xx++)
                {
                    s = s + P[int64_t(xx) * a.h + yy];
                }
                C[yy] = s;
            }
            float s = C[0];
#pragma unroll
            for (int yy = 1; yy < S; yy++)
            {
                s = s + C[yy];
            }
            out[int64_t(ch) * cells] = s * a.rq_y;
            ch++;
        }
    }
    if (!(a.magEnabled || a.histEnabled))
    {
        return;
    }
    float mn[S][S], ov[S][S];
    {
        const float* Mp = a.M + f * a.m_fs + pbase;
        const float* Sp = a.S + f * a.m_fs + pbase;
        const float* Op = a.O + f * a.m_fs + pbase;
        // a cell's S rows of one image column are S consecutive floats at an S-float-aligned offset (h % S == 0, planes
        // 256-byte aligned): one S-wide load per plane and column instead of S dword loads with the lanes 4*S bytes apart
        float mraw[S][S], sraw[S][S];
#pragma unroll
        for (int xx = 0; xx < S; xx++)
        {
            chns_load_vec<S>(Mp + int64_t(xx) * a.h, mraw[xx]);
            chns_load_vec<S>(Op + int64_t(xx) * a.h, ov[xx]);
            if (a.doNorm)
            {
                chns_load_vec<S>(Sp + int64_t(xx) * a.h, sraw[xx]);
            }
        }
#pragma unroll
        for (int xx = 0; xx < S; xx++)
        {
#pragma unroll
            for (int yy = 0; yy < S; yy++)
            {
                float m = mraw[xx][yy];
                if (a.doNorm)
                {
                    const float s = sraw[xx][yy];
                    // vector body of gradMagNorm: M * rcp(S + norm); the scalar tail
                    // (last n%4 elements) divides — n%4 == 0 here since h % shrink == 0, shrink in {2,4}... see launch
                    m = m * (1.0f / (s + a.normConst));
                }
                mn[xx][yy] = m;
                if (a.Mn)
                {
                    a.Mn[f * a.m_fs + pbase + int64_t(xx) * a.h + yy] = m;
                }
            }
        }
    }
    if (a.magEnabled)
    {
        float C[S];
#pragma unroll
        for (int yy = 0; yy < S; yy++)
        {
            float s = mn[0][yy];
#pragma unroll
            for (int xx = 1; xx < S; xx++)
            {
                s = s + mn[xx][yy];
            }
            C[yy] = s;
        }
        float s = C[0];
#pragma unroll
        for (int yy = 1; yy < S; yy++)
        {
            s = s + C[yy];
        }
        out[int64_t(ch) * cells] = s * a.rq_y;
        ch++;
    }
    if (a.histEnabled)
    {
        float H[MAXO];
#pragma unroll
        for (int b = 0; b < MAXO; b++)
        {
            H[b] = 0.f;
        }
        const float oMult = (float)a.nOrients / (a.full ? 2 * 3.14159265f : 3.14159265f);
        const float sInv2 = 1 / (float)S / (float)S;
        const int nO = a.nOrients;
#pragma unroll
        for (int xx = 0; xx < S; xx++)
        {
#pragma unroll
            for (int yy = 0; yy < S; yy++)
            {
                const float o = ov[xx][yy] * oMult;
                // (hardBin: the nearest bin takes everything; m1 = +0.0f changes no bit of the bin it is added to)
                int o0 = a.hardBin ? (int)(o + .5f) : (int)o;
                const float od = a.hardBin ? 0.f : o - (float)o0;
                if (o0 >= nO)
                {
                    o0 = 0; // o0*nb >= oMax
                }
                int o1 = o0 + 1;
                if (o1 == nO)
                {
                    o1 = 0;
                }
                const float m = mn[xx][yy] * sInv2;
                const float m1 = od * m;
                const float m0 = m - m1;
                hist_add2<MAXO>(H, o0, nO, m0, m1);
            }
        }
#pragma unroll
        for (int b = 0; b < MAXO; b++)
        {
            if (b < nO)
            {
                out[int64_t(ch + b) * cells] = H[b];
            }
        }
    }
}

// ------------------------------------------------------------------------
// imResample / resample<float> (toolbox/imResampleMex.cpp:124-383), table
// driven.  One thread per output element; the x pass value C(row) of the
// reference's column buffer is recomputed for the few source rows an output
// needs, in the reference's left-to-right association.
// ------------------------------------------------------------------------
// x-pass value C(row) for output column xb, with the column's taps already in
// scalar registers (xb is wave-uniform).
struct RsX
{
    int xa, m;      // first source column, tap count (DOWN) / 1 or 2 (UP) / k (EXACT)
    float w[4];     // first four weights (DOWN), {wt, 1-wt} (UP)
    int wofs;       // float-table offset of this column's weights (DOWN, for taps >= 4)
    bool border;    // UP: clamped column, copy
};

__device__ __forceinline__ float rs_C(int xmode, int ha, const RsX& X, const float* __restrict__ ft,
    const float* __restrict__ A, int row)
{
    if (row >= ha)
    {
        return 0.f; // C[ha .. ha+3] = 0 (:133-137)
    }
    const float* A0 = A + int64_t(X.xa) * ha + row;
    if (xmode == RS_EXACT)
    {
        float s = A0[0] + A0[ha];
        if (X.m > 2)
        {
            s = s + A0[2 * int64_t(ha)];
        }
        if (X.m > 3)
        {
            s = s + A0[3 * int64_t(ha)];
        }
        return s;
    }
    if (xmode == RS_DOWN)
    {
        float s = A0[0] * X.w[0];
        if (X.m > 1)
        {
            s = s + A0[ha] * X.w[1];
        }
        if (X.m > 2)
        {
            s = s + A0[2 * int64_t(ha)] * X.w[2];
        }
        if (X.m > 3)
        {
            s = s + A0[3 * int64_t(ha)] * X.w[3];
        }
        for (int j = 4; j < X.m; j++)
        {
            s = s + A0[int64_t(j) * ha] * ft[X.wofs + j];
        }
        return s;
    }
    if (X.border)
    {
        return A0[0];
    }
    return A0[0] * X.w[0] + A0[ha] * X.w[1];
}

// Thread layout: blockDim = (64, 4).  A wave owns 64 consecutive output rows yb
// and RS_XT consecutive output columns; the four waves of a block take adjacent
// column groups.  xb, the plane z and the level are wave-uniform, so all table
// reads for the x axis are scalar loads; each lane keeps its y taps in
// registers across the RS_XT columns.
#define RS_XT 8

__global__ void __launch_bounds__(256) k_resample(const float* __restrict__ src, float* __restrict__ dst,
    const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it, const float* __restrict__ ft, int xt)
{
    // xt: output columns per wave (the lane's y taps are reused across them); small launches use fewer for more waves
    const ResampleDesc& d = descs[blockIdx.y];
    const int ha = d.ha, hb = d.hb, wb = d.wb;
    const int ntY = (hb + 63) >> 6;
    const int ntX = (wb + 4 * xt - 1) / (4 * xt);
    int t = blockIdx.x;
    const int ytile = t % ntY;
    t /= ntY;
    const int xtile = t % ntX;
    const int z = t / ntX;
    if (z >= d.nplanes)
    {
        return;
    }
    const int yb = ytile * 64 + threadIdx.x;
    const int xb0 = __builtin_amdgcn_readfirstlane((xtile * 4 + (int)threadIdx.y) * xt);
    if (xb0 >= wb)
    {
        return;
    }
    const bool act = yb < hb;
    const int ybc = act ? yb : hb - 1;
    const int ty = z < d.c1 ? 0 : (z < d.c2 ? 1 : 2);
    const float r = d.r[ty], rk = d.rk[ty];
    const int xmode = d.xmode, ymode = d.ymode;
    const float* A = src + int64_t(blockIdx.z) * d.src_frame_stride + d.src_off + int64_t(z) * ha * d.wa;
    float* B = dst + int64_t(blockIdx.z) * d.dst_frame_stride + d.dst_off + int64_t(z) * hb * wb;

    // ---- this lane's y taps
    int ya = 0, ny = 0, q0 = 0, q1 = 0;
    float wy[4] = { 0.f, 0.f, 0.f, 0.f };
    const bool ySlow = (ymode == RS_DOWN) && d.ybd0 > 4;
    if (ymode == RS_EXACT)
    {
        ya = d.yk * ybc;
        ny = d.yk;
    }
    else if (ymode == RS_DOWN)
    {
        q0 = it[d.y_start + ybc];
        q1 = it[d.y_start + ybc + 1];
        ya = it[d.y_src + q0];
        ny = d.ybd0;
        if (!ySlow)
        {
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                if (o < ny)
                {
                    wy[o] = ft[d.y_wt + q0 + o] * r; // ywts[y] *= r (:158-161)
                }
            }
        }
    }
    else
    {
        ya = it[d.y_src + ybc];
        wy[0] = ft[d.y_wt + ybc] * r;
        wy[1] = r - wy[0];
        ny = (ybc < d.ybd0 || ybc >= hb - d.ybd1) ? 1 : 2;
    }

    for (int xi = 0; xi < xt; xi++)
    {
        const int xb = xb0 + xi;
        if (xb >= wb)
        {
            break;
        }
        // the column's x taps: one 32-byte record (host_plan.cpp) through the scalar unit instead of chasing
        // start[] -> src[] -> wt[] with dependent loads
        RsX X;
        {
            typedef uint32_t rs_u32x8 __attribute__((ext_vector_type(8)));
            typedef const __attribute__((address_space(4))) rs_u32x8* rs_cptr8;
            const rs_u32x8 xr = ((rs_cptr8)(uintptr_t)(it + d.x_col))[xb];
            X.xa = int(xr[0]);
            X.m = int(xr[1]);
            X.wofs = int(xr[2]);
            X.border = xr[3] != 0;
            X.w[0] = __uint_as_float(xr[4]);
            X.w[1] = __uint_as_float(xr[5]);
            X.w[2] = __uint_as_float(xr[6]);
            X.w[3] = __uint_as_float(xr[7]);
        }
        float v;
        if (ymode == RS_EXACT)
        {
            float s = rs_C(xmode, ha, X, ft, A, ya) + rs_C(xmode, ha, X, ft, A, ya + 1);
            if (ny > 2)
            {
                s = s + rs_C(xmode, ha, X, ft, A, ya + 2);
            }
            if (ny > 3)
            {
                s = s + rs_C(xmode, ha, X, ft, A, ya + 3);
            }
            v = s * rk;
        }
        else if (ymode == RS_DOWN)
        {
            if (!ySlow)
            {
                // U(0)+U(1)(+U(2)(+U(3))) with exactly ybd0 terms, rows ya+o (:324-348)
                v = rs_C(xmode, ha, X, ft, A, ya) * wy[0];
                v = v + rs_C(xmode, ha, X, ft, A, ya + 1) * wy[1];
                if (ny > 2)
                {
                    v = v + rs_C(xmode, ha, X, ft, A, ya + 2) * wy[2];
                }
                if (ny > 3)
                {
                    v = v + rs_C(xmode, ha, X, ft, A, ya + 3) * wy[3];
                }
            }
            else
            {
                // B0 zeroed then += over this output's entries in order (:349-356)
                v = 0.f;
                for (int q = q0; q < q1; q++)
                {
                    v = v + rs_C(xmode, ha, X, ft, A, it[d.y_src + q]) * (ft[d.y_wt + q] * r);
                }
            }
        }
        else
        {
            v = rs_C(xmode, ha, X, ft, A, ya) * wy[0];
            if (ny > 1)
            {
                v = v + rs_C(xmode, ha, X, ft, A, ya + 1) * wy[1];
            }
        }
        if (act)
        {
            B[int64_t(xb) * hb + yb] = v;
        }
    }
}

// ------------------------------------------------------------------------
// imResample, both axes UP-sampling (imResampleMex.cpp:264-280 x pass, :357-373 y pass): the frame of a model with nOctUp > 0
// (BASELINE cfg 4: 640 x 480 -> 1280 x 960).  The generic kernel above spent 3.1 ms per 192 VGA frames on it (one dword store
// per lane, a branch per tap).  Here a lane owns FOUR consecutive output rows of its column (one 16-byte store); their y taps
// are rows ya[k], ya[k] + 1 with ya[3] <= ya[0] + 3 (an up-sampled axis advances at most one source row per output row), so the
// five source rows ya[0] .. ya[0] + 4 of the two source columns are loaded once per output column, the x pass runs on those five
// rows, and the y pass picks its rows with selects on loop-invariant conditions.  Straight-line code: clamped addresses, no
// branch around a load.  Operands and order are k_resample's (rs_C, then the UP branch of the y pass): bit-identical.
// Needs hb % 4 == 0.  blockIdx.x = (plane, tile of 256 output rows, chunk of RSU_XC output columns), a wave per chunk.
// ------------------------------------------------------------------------
#define RSU_XC 32
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256) k_resample_up(const float* __restrict__ src, float* __restrict__ dst, const ResampleDesc* __restrict__ descs,
    const int32_t* __restrict__ it, const float* __restrict__ ft)
{
    const ResampleDesc& d = descs[blockIdx.y];
    const int ha = d.ha, hb = d.hb, wa = d.wa, wb = d.wb;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntY = (hb + 255) >> 8, ntX = (wb + RSU_XC - 1) / RSU_XC;
    int t = blockIdx.x * 4 + wv;
    const int xchunk = t % ntX;
    t /= ntX;
    const int ytile = t % ntY;
    const int z = t / ntY;
    if (z >= d.nplanes)
    {
        return;
    }
    const int ty = z < d.c1 ? 0 : (z < d.c2 ? 1 : 2);
    const float r = d.r[ty];
    const float* __restrict__ A = src + int64_t(blockIdx.z) * d.src_frame_stride + d.src_off + int64_t(z) * ha * wa;
    float* __restrict__ B = dst + int64_t(blockIdx.z) * d.dst_frame_stride + d.dst_off + int64_t(z) * hb * wb;
    const int yb0 = ytile * 256 + 4 * lane;
    const bool act = yb0 < hb; // (hb % 4 == 0: a lane's four rows are inside the plane together)
    int i0[4];
    float wy0[4], wy1[4];
    bool one[4];
    int rlo = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int ybc = min(yb0 + k, hb - 1);
        const int ya = it[d.y_src + ybc];
        if (k == 0)
        {
            rlo = ya;
        }
        i0[k] = min(max(ya - rlo, 0), 3);
        wy0[k] = ft[d.y_wt + ybc] * r;
        wy1[k] = r - wy0[k];
        one[k] = ybc < d.ybd0 || ybc >= hb - d.ybd1;
    }
    // the lane's five source rows: clamped addresses; rows >= ha read as 0 in the x pass's result (C[ha .. ha + 3] = 0, :133-137)
    const int r4 = min(rlo + 4, ha - 1), r03 = min(rlo, max(ha - 4, 0)); // a 4-row load that stays inside the column
    const int sh = rlo - r03;                                           // (rows rlo + i = loaded row sh + i while that is < 4)
    bool live[5];
#pragma unroll
    for (int i = 0; i < 5; i++)
    {
        live[i] = rlo + i < ha;
    }
    typedef uint32_t rs_u32x8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) rs_u32x8* rs_cptr8;
    rs_cptr8 xrec = (rs_cptr8)(uintptr_t)(it + d.x_col);
    const int xb0 = xchunk * RSU_XC, xb1 = min(xb0 + RSU_XC, wb);
    for (int xb = xb0; xb < xb1; xb++)
    {
        const rs_u32x8 xr = xrec[xb];
        const int xa = int(xr[0]);
        const bool border = xr[3] != 0;
        const float w0 = __uint_as_float(xr[4]), w1 = __uint_as_float(xr[5]);
        const float* Ac0 = A + int64_t(xa) * ha;
        const float* Ac1 = A + int64_t(min(xa + 1, wa - 1)) * ha;
        const f4u_t a4 = *reinterpret_cast<const f4u_t*>(Ac0 + r03), b4 = *reinterpret_cast<const f4u_t*>(Ac1 + r03);
        const float a5 = Ac0[r4], b5 = Ac1[r4];
        // rows rlo .. rlo + 4 from the 4-row load (shifted by sh when the column's end forced it back) and the fifth row
        float ar[5], br[5];
#pragma unroll
        for (int i = 0; i < 5; i++)
        {
            const int j = sh + i; // wave-varying, loop-invariant
            ar[i] = j == 0 ? a4.x : (j == 1 ? a4.y : (j == 2 ? a4.z : (j == 3 ? a4.w : a5)));
            br[i] = j == 0 ? b4.x : (j == 1 ? b4.y : (j == 2 ? b4.z : (j == 3 ? b4.w : b5)));
        }
        float C[5];
#pragma unroll
        for (int i = 0; i < 5; i++)
        {
            const float c = border ? ar[i] : ar[i] * w0 + br[i] * w1; // rs_C, UP
            C[i] = live[i] ? c : 0.f;
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const float c0 = i0[k] == 0 ? C[0] : (i0[k] == 1 ? C[1] : (i0[k] == 2 ? C[2] : C[3]));
            const float c1 = i0[k] == 0 ? C[1] : (i0[k] == 1 ? C[2] : (i0[k] == 2 ? C[3] : C[4]));
            const float o1 = c0 * wy0[k];
            v[k] = one[k] ? o1 : o1 + c1 * wy1[k];
        }
        if (act)
        {
            *reinterpret_cast<float4*>(B + int64_t(xb) * hb + yb0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ------------------------------------------------------------------------
// Threshold-rank cells (host_plan.h): rank(v) = number of the channel's distinct node thresholds <= v, the 16-bit cell
// the cascade tile kernel reads.  `rec` = the CHANNEL's bucket records in LDS: one 16-byte read per cell.
// ------------------------------------------------------------------------
struct RankFn
{
    int32_t shift, base, nbm1;
    uint32_t mask;
};
// the read: issue it early, count late (rank_count) — the level kernels put a column's other LDS traffic in between
__device__ __forceinline__ uint4 rank_fetch(float v, const RankFn& f, const uint4* rec, uint32_t& low)
{
    // (a negative v lands in bucket 0 through the arithmetic shift; its rank is forced to 0 in rank_count)
    const int key = __float_as_int(v);
    int b; // clamp to [0, nbm1] in one instruction (the compiler cannot prove nbm1 >= 0 and keeps v_max + v_min)
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(b) : "v"((key >> f.shift) - f.base), "v"(f.nbm1));
    low = (uint32_t(key) & f.mask) | 0x8000u; // the guard bit of rank_count's packed compares
    return rec[b];
}
__device__ __forceinline__ uint32_t rank_count(float v, const uint4& r, uint32_t low)
{
    static_assert(RANK_WINDOW == 7 && sizeof(RankRec) == 16 && RANK_UNUSED == 0x8000, "record layout");
    // Seven `t <= low` as four subtractions: low keys are < 0x8000 and carry the guard bit 0x8000 here, slots hold
    // <= 0x8000, so (low | 0x8000) - t has bit 15 set exactly when t <= low and never borrows from the upper half.  The
    // indicator bits (15 and 31 of every difference; dword 0's upper half is `lo`, not a slot) are moved to distinct
    // positions and counted with one v_bcnt that also adds `lo`.  18 VALU at full rate; the seven v_cmp_le_u32_sdwa +
    // v_addc/v_cndmask of the scalar form issue at half rate on gfx950 (profiles/ubench/valu_rate.hip).
    const uint32_t X = low | (low << 16);
    const uint32_t m0 = (X - r.x) & 0x00008000u;
    const uint32_t m1 = (X - r.y) & 0x80008000u;
    const uint32_t m2 = (X - r.z) & 0x80008000u;
    const uint32_t m3 = (X - r.w) & 0x80008000u;
    const uint32_t n = uint32_t(__builtin_popcount(m0 | (m1 >> 1) | (m2 >> 2) | (m3 >> 3))) + (r.x >> 16);
    return v < 0.f ? 0u : n; // every threshold is >= 0 (buildRankTables): a negative cell is below all of them
}
__device__ __forceinline__ uint32_t rank_cell(float v, const RankFn& f, const uint4* rec)
{
    uint32_t low;
    const uint4 r = rank_fetch(v, f, rec, low);
    return rank_count(v, r, low);
}

// copy a channel's records into LDS; returns the bucket function
__device__ __forceinline__ RankFn rank_tables_to_lds(const RankChan& rc, const RankRec* __restrict__ recG, uint4* recL, int tid, int nThreads)
{
    const uint4* src = reinterpret_cast<const uint4*>(recG + rc.recOff);
    for (int i = tid; i < rc.nb; i += nThreads)
    {
        recL[i] = src[i];
    }
    RankFn f;
    f.shift = rc.shift;
    f.base = rc.base;
    f.nbm1 = rc.nb - 1;
    f.mask = (1u << rc.shift) - 1u;
    return f;
}

// Stand-alone form: the fused float pyramid -> rank cells, one workgroup per (64-column chunk, level x channel, frame).
// Used when the level kernels that emit rank cells themselves do not cover the plan (and by the parity tests of both).
struct RankJob
{
    int64_t src_off; // float offset of the level in one frame's fused pyramid
    int64_t dst_off; // cell offset of the level in one frame's rank pyramid
    int32_t hP, wP, pitchR, pad_;
};
constexpr int RANK_CHUNK_COLS = 64;
__global__ void __launch_bounds__(256) k_rank(const float* __restrict__ pyr, int64_t pyr_fs, uint16_t* __restrict__ out, int64_t out_fs,
    const RankJob* __restrict__ jobs, int nChns, const RankChan* __restrict__ chan, const RankRec* __restrict__ recG)
{
    extern __shared__ uint4 ldsRec[];
    const int lvl = blockIdx.y / nChns, z = blockIdx.y - lvl * nChns;
    const RankJob J = jobs[lvl];
    const int c0 = blockIdx.x * RANK_CHUNK_COLS;
    if (c0 >= J.wP)
    {
        return;
    }
    const RankFn fn = rank_tables_to_lds(chan[z], recG, ldsRec, threadIdx.x, 256);
    __syncthreads();
    const float* __restrict__ src = pyr + int64_t(blockIdx.z) * pyr_fs + J.src_off + int64_t(z) * J.hP * J.wP;
    uint16_t* __restrict__ dst = out + int64_t(blockIdx.z) * out_fs + J.dst_off + int64_t(z) * J.pitchR * J.wP;
    const int c1 = min(c0 + RANK_CHUNK_COLS, J.wP);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = c0 + wv; c < c1; c += 4)
    {
        for (int r = lane; r < J.hP; r += 64)
        {
            dst[int64_t(c) * J.pitchR + r] = uint16_t(rank_cell(src[int64_t(c) * J.hP + r], fn, ldsRec));
        }
    }
}

// ------------------------------------------------------------------------
// Fused level kernel: approximated-scale resample (chnsPyramid.cpp:385-397) +
// final convTri1 smoothing with the in-place aliasing (:399-407) + placement in
// the padded, fused pyramid (:410-435), one pass, no intermediate plane.
//
// ONE WAVE owns one (level, channel) plane.  Lane l holds rows l, l+64, ... (R
// per lane), so a column is R coalesced 256-byte accesses.  The smoothing
// recursion along image-x needs T[y-1] and T[y+1] every column: with the rows
// interleaved by 64 those live in the neighbouring LANES of the same register,
// fetched with DPP wave rotates (lane 0 / 63 take the wrap-around value from the
// adjacent register) — no LDS, no barrier, so a CU runs as many planes
// concurrently as it has wave slots instead of one barrier-synchronised
// workgroup per plane.  The column fed to the recursion is produced on the fly:
// for a real level it is read from the raw channels, for an approximated level it
// is resampled from the real level's raw channels with exactly the arithmetic of
// k_resample (x pass then y pass, reference association order); the column two
// steps ahead is requested while the current one is filtered.
// ------------------------------------------------------------------------
struct LevelJob
{
    int32_t hC, wC, out_cs, desc; // desc: index of the ResampleDesc of an approximated level, -1 for a real level
    int32_t kind, pad_;           // R * 8 + mode: the specialisation k_level_all dispatches to
    int64_t in_off;               // real level: float offset of its raw channels in the per-frame channel buffer
    int64_t raw_off;              // where the level's raw (unsmoothed) channels go when taps are kept
    int64_t out_off;              // float offset of the level's interior in the per-frame pyramid
    int64_t in_ps, out_ps;        // plane strides
    int64_t rank_off, rank_ps;    // rank pyramid (16-bit cells): cell offset of the level's interior in one frame, plane stride
    int32_t rank_cs, pad2_;       // rank pyramid: cells between columns
};


typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8* cptr8_t;

// LM_*: which column source a wave uses.  Levels are launched in groups of equal
// (R, mode), so both are compile-time: R = ceil(h/64) exactly (every register but
// the last holds 64 valid rows: no per-register guards) and only the arithmetic of
// one resampling mode is in the instruction stream.  The code is written
// branch-free on purpose — loads are unconditional on clamped addresses and
// conditions are applied with selects afterwards — because a conditional load
// becomes its own basic block with a full memory wait, which serialises every
// access of the column (measured: 5x slower than the barrier kernel it replaces).
enum
{
    LM_REAL = 0, // raw channels of a real level
    LM_DD = 1,   // approximated level: x down, y down
    LM_DU = 2,   // x down, y up
    LM_UD = 3,   // x up, y down
    LM_UU = 4
};

// Buffer addressing (SRD in SGPRs + 32-bit per-lane byte offset + scalar byte offset): every access of the level
// kernel is `buffer_load/store v, voff, srd, soff offen`, with no per-access VALU address arithmetic.  The
// descriptor inputs are made provably wave-uniform with readfirstlane (cdna_hip_programming.md T20).
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, int64_t bytes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(a)), hi = __builtin_amdgcn_readfirstlane(uint32_t(a >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(uint32_t(bytes > 0xffffffffll ? 0xffffffffll : bytes));
    void* p = reinterpret_cast<void*>((uint64_t(hi) << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, int(n), 0x00020000);
}
__device__ __forceinline__ float buf_ld(srd_t r, uint32_t voff, uint32_t soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st(srd_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// Source columns of an approximated level.  The x pass of imResample combines JX adjacent source columns per output
// column and successive output columns move on by s = 0, 1 or 2 source columns.  Every source column is fetched ONCE,
// by LDS-DMA (`buffer_load_dword ... lds`, lanes on SOURCE rows), into a wave-private ring of NB column slots, LA = NB - JX
// columns ahead of the window; the x pass reads its JX columns from the ring, runs once per source row, and hands the
// column to the y pass through the wave's column buffer (row r at float r, gathered by each output row's y taps).
// Round 2 measured what the register window of round 1 cost: its new columns were requested ONE step ahead, inside
// wave-uniform branches on the shift, so every step of every approximated plane (27 of 31 levels) waited a full memory
// round trip (4.7k cycles per column step with 14 waves per CU: 2.4 TB/s); deeper register prefetch would need copies
// of registers that loads are still writing.  A ring has no copies, and its wait is a constant: when the window needs
// columns <= c, columns <= c + LA have been requested, so at least LA * RS younger requests exist and
// `s_waitcnt vmcnt(LA * RS)` (completion is in order; stores only add younger entries) covers c.
// Per element the operands and their order are those of k_resample (x pass then y pass): bit-identical.
// waves (= channel planes) per workgroup of the level kernels: 2, so that ten channels are 5 full workgroups
// (4 since round 3 — a multiple of the CU's four SIMDs: six measured 12 % slower, two SIMDs carry twice the waves —: the
// waves of a workgroup take the SAME channel of four frames, so that one copy of the channel's
// threshold-rank tables in LDS serves the workgroup)
#ifndef ACF_LEVEL_WAVES
#define ACF_LEVEL_WAVES 4
#endif
constexpr int LEVEL_WAVES = ACF_LEVEL_WAVES;
constexpr int LEVEL_RING_FLOATS = 2304; // 9 KB per wave: 6 slots of 6 x 64 rows ... 12 slots of <= 3 x 64 rows
// ... and 7.5 KB (5 slots of 6 x 64 rows) in the forms that also hold a channel's rank records in LDS: a workgroup's LDS is
// allocated in 1280-byte granules and three workgroups per CU must fit (4 x 10.75 KB + 10.6 KB of records was 43 granules:
// two workgroups per CU, k_level 1.25 -> 2.0 ms); one column of look-ahead less costs the seven largest levels ~6 %
constexpr int LEVEL_RING_FLOATS_RANK = 1920;

template <int R, int MODE, int RING = LEVEL_RING_FLOATS>
struct LevelWindow
{
    static constexpr bool XDOWN = MODE == LM_DD || MODE == LM_DU;
    static constexpr bool YDOWN = MODE == LM_DD || MODE == LM_UD;
    static constexpr int NY = YDOWN ? 3 : 2;
    static constexpr int JX = XDOWN ? 3 : 2;
    static constexpr int RS = YDOWN ? (3 * R + 1) / 2 : R; // source rows per lane: ha <= 64 * RS (host-checked)
    static constexpr int NBRAW = RING / (RS * 64);
    static constexpr int NB = NBRAW > 12 ? 12 : (NBRAW < JX + 2 ? JX + 2 : NBRAW); // ring slots
    static constexpr int LA = NB - JX;                                             // columns requested ahead of the window
    static constexpr int LDS_FLOATS = RS * 64 * (NB + 1);                          // x-pass column + ring, per wave
    static_assert(LA * RS <= 63, "vmcnt immediate");

    uint32_t srow[RS]; // byte offset of the lane's (clamped) source rows in a column
    int issued;        // last source column requested (wave-uniform)
    float* ring;       // NB slots of RS * 64 floats: column c in slot c % NB, source row q at float q

    __device__ __forceinline__ void request(srd_t A, int col, int ha, int wa) const
    {
        const uint32_t cb = uint32_t(min(col, wa - 1)) * uint32_t(ha) * 4u;
        float* slot = ring + (uint32_t(col) % uint32_t(NB)) * (RS * 64);
#pragma unroll
        for (int r = 0; r < RS; r++)
        {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(A, (lptr_t)(slot + 64 * r), 4, srow[r], cb, 0, 0);
        }
    }

    __device__ __forceinline__ void init(srd_t A, float* ring_, int lane, int ha, int wa, int xa0)
    {
#pragma unroll
        for (int r = 0; r < RS; r++)
        {
            // rows >= ha: an offset beyond the descriptor's range — the load returns 0, so the ring, the x pass's column
            // (0 * w) and with it the column buffer hold +0 there: the zeroed tail of the reference's column buffer
            // (imResampleMex.cpp:133-137), which the y taps of the plane's last rows read (level_column)
            srow[r] = lane + 64 * r < ha ? 4u * uint32_t(lane + 64 * r) : 0x40000000u;
        }
        ring = ring_;
        for (int k = 0; k < JX + LA; k++)
        {
            request(A, xa0 + k, ha, wa);
        }
        issued = xa0 + JX + LA - 1;
    }

    // the window moves to source column xa (wave-uniform): request the columns that enter the look-ahead
    __device__ __forceinline__ void advance(srd_t A, int xa, int ha, int wa)
    {
        const int target = xa + JX - 1 + LA;
        while (issued < target) // 0, 1 or 2 rounds for ratios within 2^(+-1/2)
        {
            issued++;
            request(A, issued, ha, wa);
        }
    }

    // columns xa .. xa + JX - 1 of the lane's source rows
    __device__ __forceinline__ void read(float (&d)[JX][RS], int xa, int lane) const
    {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA * RS) : "memory");
#pragma unroll
        for (int j = 0; j < JX; j++)
        {
            const float* slot = ring + (uint32_t(xa + j) % uint32_t(NB)) * (RS * 64) + lane;
#pragma unroll
            for (int r = 0; r < RS; r++)
            {
                d[j][r] = slot[64 * r];
            }
        }
    }
};

template <int R>
struct LaneTaps
{
    uint32_t roff[R][3]; // clamped source row of y tap o, as a BYTE offset in a source column
    float wy[R][3];      // y weights, gain folded in (imResampleMex.cpp:158-161)
    uint32_t one[R];     // y up: a clamped border row uses one tap only
};

// One column of an approximated level: x pass then y pass, reference association order
// (imResampleMex.cpp:198-280, 319-373).
template <int R, int MODE, int RING>
__device__ __forceinline__ void level_column(float (&v)[R], int xb, srd_t A, int h, const uint32_t (&yoff)[R],
    int ha, int wa, int ny, const u32x8& xr, const LaneTaps<R>& tp, srd_t raw, bool haveRaw, bool lastOk,
    LevelWindow<R, MODE == LM_REAL ? LM_DD : MODE, RING>& win, float* ldsCol, int lane)
{
    if (MODE == LM_REAL)
    {
        const uint32_t col = uint32_t(xb) * uint32_t(h) * 4u; // planes are < 2^30 floats: byte offsets fit 32 bits
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            v[k] = buf_ld(A, yoff[k], col);
        }
        return;
    }
    constexpr bool XDOWN = MODE == LM_DD || MODE == LM_DU;
    constexpr bool YDOWN = MODE == LM_DD || MODE == LM_UD;
    // An approximated level is within a factor 2^(+-1/2) of its real level, so an output has at most three
    // taps per axis when down-sampling and two when up-sampling (the plan falls back to separate launches
    // otherwise).  All JX*NY source values of every register are requested before any is used.
    typedef LevelWindow<R, MODE == LM_REAL ? LM_DD : MODE, RING> Win;
    constexpr int NY = Win::NY;
    constexpr int JX = Win::JX;
    const int xa = int(xr[0]), m = int(xr[1]);
    const bool border = xr[3] != 0;
    float w[4] = { __uint_as_float(xr[4]), __uint_as_float(xr[5]), __uint_as_float(xr[6]), __uint_as_float(xr[7]) };
    constexpr int RS = Win::RS;
    if (RS >= 3 && RS < 6)
    {
        // the column's x weights multiply RS rows each: three v_mov from SGPRs are cheaper than RS * JX SGPR operands
        ACF_PIN_V(w[0]);
        ACF_PIN_V(w[1]);
        ACF_PIN_V(w[2]);
    }
    win.advance(A, xa, ha, wa);
    float sc[JX][RS];
    win.read(sc, xa, lane);
    // x pass on the lane's source rows: taps accumulate left to right (imResampleMex.cpp:198-280).  The tap count m and
    // the border flag are wave-uniform per column: branch on them once, around arithmetic only.
    float Cr[RS];
    if (XDOWN)
    {
        if (m == 2)
        {
#pragma unroll
            for (int r = 0; r < RS; r++)
            {
                Cr[r] = sc[0][r] * w[0] + sc[1][r] * w[1];
            }
        }
        else if (m >= 3)
        {
#pragma unroll
            for (int r = 0; r < RS; r++)
            {
                Cr[r] = sc[0][r] * w[0] + sc[1][r] * w[1] + sc[JX - 1][r] * w[2];
            }
        }
        else
        {
#pragma unroll
            for (int r = 0; r < RS; r++)
            {
                Cr[r] = sc[0][r] * w[0];
            }
        }
    }
    else if (border)
    {
        // up: a clamped border column is copied (:264-280)
#pragma unroll
        for (int r = 0; r < RS; r++)
        {
            Cr[r] = sc[0][r];
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < RS; r++)
        {
            Cr[r] = sc[0][r] * w[0] + sc[1][r] * w[1]; // A0*wt + A1*(1-wt)
        }
    }
    // hand the column to the y pass: source row q at ldsCol[q]; an output row gathers its NY taps (byte offsets
    // tp.roff).  One wave owns the buffer and LDS operations of a wave complete in order, so the writes below cannot
    // pass the previous step's gathers and the gathers cannot pass the writes.
#pragma unroll
    for (int r = 0; r < RS; r++)
    {
        ldsCol[lane + 64 * r] = Cr[r];
    }
    float C[R][NY];
#pragma unroll
    for (int k = 0; k < R; k++)
    {
#pragma unroll
        for (int o = 0; o < NY; o++)
        {
            C[k][o] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldsCol) + tp.roff[k][o]);
        }
    }
    // (a tap row >= ha reads the zeroed tail of the reference's column buffer, :133-137: rows ha .. 64 * RS - 1 of ldsCol
    // are +0 — LevelWindow::init — and tp.roff points there: no select.  Round 2 applied `bad` masks here: 12 v_cndmask
    // per column step on SGPR masks that no longer fitted the SGPR file, i.e. 24 v_readlane of spilled masks as well.)
    // y pass (:319-373)
    if (YDOWN)
    {
        // U(0)+U(1)(+U(2)) with exactly ybd0 terms (:324-348); ny is wave-uniform
        if (ny > 2)
        {
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                v[k] = C[k][0] * tp.wy[k][0] + C[k][1] * tp.wy[k][1] + C[k][2] * tp.wy[k][2];
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                v[k] = C[k][0] * tp.wy[k][0] + C[k][1] * tp.wy[k][1];
            }
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            const float o1 = C[k][0] * tp.wy[k][0];
            const float o2 = o1 + C[k][1] * tp.wy[k][1];
            v[k] = tp.one[k] ? o1 : o2;
        }
    }
    if (haveRaw)
    {
        const uint32_t rc = uint32_t(xb) * uint32_t(h) * 4u;
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            if (k < R - 1 || lastOk)
            {
                buf_st(raw, yoff[k], rc, v[k]);
            }
        }
    }
}

// What a level kernel writes: the float level (the Pyramid the API returns), its threshold-rank cells (what the cascade's
// tile kernel reads), or both.
enum
{
    LO_F32 = 1,
    LO_RANK = 2
};
struct LevelRank
{
    uint16_t* out;        // rank pyramid
    int64_t fs;           // cells per frame
    RankFn fn;            // the channel's bucket function
    const uint4* rec;     // the channel's bucket records, in LDS
};

// A plane's column chain cut into speculative segments (SEG = 1; the smoothing's recursion is k_smooth_vec's: see
// "speculative segments" there): columns [x0, x1), started `x0 - xs` columns early from the border formula; the state
// after the warm-up goes to `spec`, the state after the last column to `tru` (the next segment's slot), both [hC] floats;
// k_level_verify compares them and a repair launch recomputes the planes that differ as one chain.  For the small batches
// where a level's 480-step chain is the launch's duration (one frame: 368 -> ~100 us).
struct LevelSeg
{
    int x0, x1, xs;
    float* spec; // nullptr: first segment
    float* tru;  // nullptr: last segment
};

template <int R, int MODE, int OUT, int SEG>
__device__ __forceinline__ void level_body(const LevelJob& J, const int64_t f, const float* __restrict__ chns, float* __restrict__ pyr, float* __restrict__ rawOut,
    const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it, const float* __restrict__ ft,
    int nChns, int64_t chns_fs, int64_t pyr_fs, float p, float* __restrict__ dump, float* ldsBlock, int ldsWaveFloats, const LevelRank& rk,
    const LevelSeg& sg)
{
    // the plane index is the same for the 64 lanes of a wave (and of the workgroup: its waves are the same plane of
    // LEVEL_WAVES frames); say so (readfirstlane), or every plane pointer is treated as per-lane and all address
    // arithmetic lands on the VALU in 64 bits
    const int z = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const int lane = threadIdx.x & 63;
    const int h = J.hC, w = J.wC;
    const srd_t Osrd = make_srd(pyr + f * pyr_fs + J.out_off + int64_t(z) * J.out_ps, (int64_t(w - 1) * J.out_cs + h) * 4);
    // rank cells leave two rows per lane (below): the descriptor starts `par` cells before the level's interior so that
    // it starts on a 4-byte boundary (pitches and plane sizes are multiples of 8 cells), and reaches one cell past the
    // interior's last row
    const int par = int(J.rank_off & 1);
    const srd_t Rsrd = (OUT & LO_RANK) ? make_srd(rk.out + f * rk.fs + (J.rank_off - par) + int64_t(z) * J.rank_ps, (int64_t(w - 1) * J.rank_cs + h + 1 + par) * 2) : Osrd;
    srd_t A, raw = Osrd;
    bool haveRaw = false;
    int ha = 0, wa = 0, ny = 0;
    LaneTaps<R> tp;
    uint32_t yoff[R];
#pragma unroll
    for (int k = 0; k < R; k++)
    {
        yoff[k] = 4u * uint32_t(min(lane + 64 * k, h - 1)); // byte offset of the lane's (clamped) row in a column
    }
    const bool lastOk = lane + 64 * (R - 1) < h; // rows of the last register beyond the plane are clamped duplicates: never stored
    const int32_t* xcol = it;
    if (MODE == LM_REAL)
    {
        A = make_srd(chns + f * chns_fs + J.in_off + int64_t(z) * J.in_ps, J.in_ps * 4);
    }
    else
    {
        const ResampleDesc& d = descs[J.desc];
        A = make_srd(chns + f * chns_fs + d.src_off + int64_t(z) * d.ha * d.wa, int64_t(d.ha) * d.wa * 4);
        haveRaw = rawOut != nullptr;
        if (haveRaw)
        {
            raw = make_srd(rawOut + f * chns_fs + J.raw_off + int64_t(z) * J.in_ps, J.in_ps * 4);
        }
        const int ty = z < d.c1 ? 0 : (z < d.c2 ? 1 : 2);
        const float r = d.r[ty];
        const int hb = d.hb;
        ha = d.ha;
        wa = d.wa;
        xcol = it + d.x_col;
        constexpr bool YDOWN = MODE == LM_DD || MODE == LM_UD;
        typedef LevelWindow<R, MODE == LM_REAL ? LM_DD : MODE> Win0; // (RS does not depend on the ring size)
        ny = YDOWN ? d.ybd0 : 2;
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            const int ybc = int(yoff[k] >> 2); // == min(yb, hb - 1)
            int ya;
            tp.wy[k][0] = tp.wy[k][1] = tp.wy[k][2] = 0.f;
            tp.one[k] = 0;
            if (YDOWN)
            {
                const int q0 = it[d.y_start + ybc];
                ya = it[d.y_src + q0];
#pragma unroll
                for (int o = 0; o < 3; o++)
                {
                    if (o < d.ybd0)
                    {
                        tp.wy[k][o] = ft[d.y_wt + q0 + o] * r;
                    }
                }
            }
            else
            {
                ya = it[d.y_src + ybc];
                tp.wy[k][0] = ft[d.y_wt + ybc] * r;
                tp.wy[k][1] = r - tp.wy[k][0];
                tp.one[k] = (ybc < d.ybd0 || ybc >= hb - d.ybd1) ? 1u : 0u;
            }
#pragma unroll
            for (int o = 0; o < 3; o++)
            {
                // rows ha .. 64 * RS - 1 of the column buffer hold +0 (the host keeps ha < 64 * RS for this kernel)
                tp.roff[k][o] = 4u * uint32_t(min(ya + o, 64 * Win0::RS - 1));
            }
        }
    }
    // (nrm, p, p1 in VGPRs: a VALU instruction with an SGPR operand issues at 1.7x the cost of one without on gfx950,
    // profiles/ubench/valu_rate.hip, and these are operands of ~5 multiplies per cell)
    float nrm = 1.0f / ((p + 2) * (p + 2));
    const float p1 = 1 + p;
    float pv = p;
    if (R <= 5)
    {
        // (p1 only multiplies in the two border rows' registers: it stays scalar; R >= 6 has no VGPR to spare)
        ACF_PIN_V(nrm);
        ACF_PIN_V(pv);
    }
    // Column records come through the scalar unit (constant address space) ahead of the column loads
    // they drive; column loads are issued three steps before the recursion consumes them.  Four column
    // buffers rotate through the roles (loop unrolled 4x) so no register is copied — a copy would force
    // the wait for the load that feeds it.  The loop starts at i = -3: the first three steps only load.
    cptr8_t xrec = (cptr8_t)(uintptr_t)xcol;
    const u32x8 zrec = { 0, 0, 0, 0, 0, 0, 0, 0 };
    float b0[R], b1[R], b2[R], b3[R], prev[R];
#pragma unroll
    for (int k = 0; k < R; k++)
    {
        b0[k] = b1[k] = b2[k] = b3[k] = prev[k] = 0.f;
    }
    // The main loop body is straight-line code (no branch, not even a wave-uniform one): column indices
    // are clamped instead of guarded and the last register's rows beyond the plane are stored to a dump
    // slot instead of being masked.  vmcnt completes in order, and the compiler can only leave older
    // loads in flight across a step if it sees the whole step as one basic block — with a branch per
    // store it waited for the loads it had just issued, every step.
    const int xs = SEG ? sg.xs : 0, x0 = SEG ? sg.x0 : 0, x1 = SEG ? sg.x1 : w; // this wave's columns (SEG = 0: the whole plane)
    u32x8 xr = (MODE == LM_REAL) ? zrec : xrec[xs];
    u32x8 xrn = (MODE == LM_REAL) ? zrec : xrec[min(xs + 1, w - 1)];
    // approximated levels: the source-column window (registers) and the wave's x-pass column buffer (LDS)
    constexpr int RING = (OUT & LO_RANK) ? LEVEL_RING_FLOATS_RANK : LEVEL_RING_FLOATS;
    typedef LevelWindow<R, MODE == LM_REAL ? LM_DD : MODE, RING> Win;
    Win lw;
    float* ldsCol = ldsBlock + (threadIdx.x >> 6) * ldsWaveFloats; // the wave's x-pass column, then its ring
    if (MODE != LM_REAL)
    {
        lw.init(A, ldsCol + Win::RS * 64, lane, ha, wa, int(xr[0]));
    }
    const int lastLane = (h - 1) & 63; // lane holding row h-1 in the last register
#define LV_LOAD(FAR, COL)                                                                                           \
    {                                                                                                               \
        level_column<R, MODE, RING>(FAR, min((COL), w - 1), A, h, yoff, ha, wa, ny, xr, tp, raw, haveRaw, lastOk, lw, ldsCol, lane); \
        xr = xrn;                                                                                                   \
        xrn = (MODE == LM_REAL) ? zrec : xrec[min((COL) + 2, w - 1)];                                               \
    }
#define LV_FILTER(I, CUR, NXT)                                                                                      \
    {                                                                                                               \
        const int i_ = (I);                                                                                         \
        float T[R], up[R], dn[R];                                                                                   \
        _Pragma("unroll") for (int k = 0; k < R; k++)                                                               \
        {                                                                                                           \
            const float Im = CUR[k];                                                                                \
            const float Ir = (i_ < w - 1) ? NXT[k] : Im;                                                            \
            const float Il = (i_ == xs) ? Im : prev[k]; /* Il = Im at i == 0 (convConst.cpp:503-507); a segment's warm-up starts the same way */ \
            T[k] = nrm * (Il + pv * Im + Ir);                                                                       \
            up[k] = wave_ror1(T[k]); /* T[y-1] for lanes 1..63 */                                                   \
            dn[k] = wave_rol1(T[k]); /* T[y+1] for lanes 0..62 */                                                   \
        }                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < R; k++)                                                               \
        {                                                                                                           \
            const int y = lane + 64 * k;                                                                            \
            const float tm = (lane == 0) ? up[k > 0 ? k - 1 : 0] : up[k];       /* row y-1 */                       \
            const float tpv = (lane == 63) ? dn[k + 1 < R ? k + 1 : k] : dn[k]; /* row y+1 */                       \
            const float mid = tm + pv * T[k] + tpv;                                                                 \
            /* row 0 is lane 0 of register 0 and row h - 1 lives in the last register (R == ceil(h / 64)): the other */ \
            /* registers have neither border form nor select                                                      */ \
            float o = mid;                                                                                          \
            if (k == R - 1)                                                                                         \
            {                                                                                                       \
                o = (y == h - 1) ? tm + p1 * T[k] : o;                                                              \
            }                                                                                                       \
            if (k == 0)                                                                                             \
            {                                                                                                       \
                o = (y == 0) ? p1 * T[k] + tpv : o;                                                                 \
            }                                                                                                       \
            prev[k] = o;                                                                                            \
            /* lanes past the end of the plane (last register only) are clamped to row h-1: they store row h-1's */ \
            /* value to row h-1's address, so every store is unconditional and base + 32-bit offset             */ \
            const float ov = (k < R - 1 || lastOk) ? o : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o), lastLane)); \
            if (OUT & LO_F32)                                                                                       \
            {                                                                                                       \
                /* (warm-up columns of a segment: the store goes out of the descriptor's range and is dropped) */   \
                buf_st(Osrd, yoff[k], (!SEG || i_ >= x0) ? uint32_t(i_) * uint32_t(J.out_cs) * 4u : 0x40000000u, ov); \
            }                                                                                                       \
        }                                                                                                           \
    }
    // (storing four columns at a time instead of one was measured: no gain, 16 more VGPRs)
    // Rank cells one column behind: `prev` holds column i - 1's output during step i.  Its bucket records are requested
    // BEFORE the next column's LDS traffic (ring reads, y-tap gathers) and counted after it, so the one LDS round trip of
    // a rank hides behind waits the step has anyway (LDS operations of a wave complete in order); emitting the rank
    // inline, right after the float store, cost a dependent round trip per column step: 1.27 -> 1.9 ms per 96 frames.
    uint4 rrec[R];
    uint32_t rlow[R];
    // 2-byte stores are slow (one 128-byte buffer_store_short per register measured as much as 1.5 float stores: the rank
    // cells alone made the kernel 28 % slower than the floats alone).  Rows y - 1 and y leave as ONE dword from the lane of
    // row y, for the rows with (y + par) odd — i.e. cell pairs that start on 4 bytes —, row y - 1's rank coming from the
    // neighbouring lane; the other lanes' stores go out of the descriptor's range and are dropped.  Pairs may cover one
    // cell above the interior (y = 0, par = 1) or below it (y = h): a border cell k_pad_reflect rewrites afterwards, or
    // the unused tail of the column's pitch.  (hC a multiple of 64 with par = 1 has no lane for y = h: the host keeps such
    // plans on k_rank.)
    uint32_t rvoff[R];
    if (OUT & LO_RANK)
    {
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            const int y = lane + 64 * k;
            rvoff[k] = (((y + par) & 1) && y <= h) ? uint32_t(2 * (y - 1 + par)) : 0x40000000u;
        }
    }
#define LV_RANK_FETCH()                                                                                             \
    if (OUT & LO_RANK)                                                                                              \
    {                                                                                                               \
        _Pragma("unroll") for (int k = 0; k < R; k++)                                                               \
        {                                                                                                           \
            rrec[k] = rank_fetch(prev[k], rk.fn, rk.rec, rlow[k]);                                                  \
        }                                                                                                           \
    }
    // column COL of the rank level <- ranks of `prev` (COL < 0: nothing yet; the stores go out of the descriptor's range
    // and are dropped by the buffer bounds check: no branch)
#define LV_RANK_STORE(COL)                                                                                          \
    if (OUT & LO_RANK)                                                                                              \
    {                                                                                                               \
        const int c_ = (COL);                                                                                       \
        const uint32_t so_ = c_ >= x0 ? uint32_t(c_) * uint32_t(J.rank_cs) * 2u : 0x40000000u;                      \
        uint32_t n_[R], up_[R];                                                                                     \
        _Pragma("unroll") for (int k = 0; k < R; k++)                                                               \
        {                                                                                                           \
            n_[k] = rank_count(prev[k], rrec[k], rlow[k]);                                                          \
            up_[k] = __float_as_uint(wave_ror1(__uint_as_float(n_[k]))); /* row y-1's rank for lanes 1..63 */       \
        }                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < R; k++)                                                               \
        {                                                                                                           \
            const uint32_t lo_ = (lane == 0) ? up_[k > 0 ? k - 1 : 0] : up_[k];                                     \
            __builtin_amdgcn_raw_buffer_store_b32((n_[k] << 16) | (lo_ & 0xffffu), Rsrd, rvoff[k], so_, 0);         \
        }                                                                                                           \
    }
    // prologue: columns xs, xs + 1, xs + 2 -> b0, b1, b2
    LV_LOAD(b0, xs);
    LV_LOAD(b1, xs + 1);
    LV_LOAD(b2, xs + 2);
    // a segment's state after its warm-up (x0 - xs is a multiple of 4: the loop below passes i == x0)
    const srd_t Ssrd = (SEG && sg.spec) ? make_srd(sg.spec, int64_t(h) * 4) : Osrd;
    int i = xs;
    for (; i + 3 < x1; i += 4)
    {
        if (SEG)
        {
            const uint32_t ss_ = (sg.spec && i == x0) ? 0u : 0x40000000u;
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                buf_st(Ssrd, yoff[k], ss_, prev[k]);
            }
        }
        LV_RANK_FETCH();
        LV_LOAD(b3, i + 3);
        LV_RANK_STORE(i - 1);
        LV_FILTER(i, b0, b1);
        LV_RANK_FETCH();
        LV_LOAD(b0, i + 4);
        LV_RANK_STORE(i);
        LV_FILTER(i + 1, b1, b2);
        LV_RANK_FETCH();
        LV_LOAD(b1, i + 5);
        LV_RANK_STORE(i + 1);
        LV_FILTER(i + 2, b2, b3);
        LV_RANK_FETCH();
        LV_LOAD(b2, i + 6);
        LV_RANK_STORE(i + 2);
        LV_FILTER(i + 3, b3, b0);
    }
    if (SEG && sg.spec && i == x0)
    {
        // (a last segment shorter than four columns: the loop above never reached x0)
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            buf_st(Ssrd, yoff[k], 0u, prev[k]);
        }
    }
    // tail: up to three columns; their inputs are already in b0, b1, b2
    if (i < x1)
    {
        LV_RANK_FETCH();
        LV_RANK_STORE(i - 1);
        LV_FILTER(i, b0, b1);
    }
    if (i + 1 < x1)
    {
        LV_RANK_FETCH();
        LV_RANK_STORE(i);
        LV_FILTER(i + 1, b1, b2);
    }
    if (i + 2 < x1)
    {
        LV_RANK_FETCH();
        LV_RANK_STORE(i + 1);
        LV_FILTER(i + 2, b2, b2);
    }
    LV_RANK_FETCH();
    LV_RANK_STORE(x1 - 1);
    if (SEG && sg.tru)
    {
        // the state the next segment's warm-up must have reached (clamped duplicate lanes write their own row again)
        const srd_t Tsrd = make_srd(sg.tru, int64_t(h) * 4);
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            buf_st(Tsrd, yoff[k], 0u, prev[k]);
        }
    }
#undef LV_RANK_FETCH
#undef LV_RANK_STORE
#undef LV_LOAD
#undef LV_FILTER
}

// One launch per run of levels with equal (R, mode): blockIdx.x = channel, blockIdx.y = level of the run, blockIdx.z =
// group of LEVEL_WAVES frames (wave w of the workgroup takes frame blockIdx.z * LEVEL_WAVES + w).  Float output only: the
// runs that do not fit k_level_all are the large planes of large frames; their rank cells come from k_rank.
template <int R, int MODE>
constexpr int levelWaveFloats()
{
    return MODE == LM_REAL ? 1 : LevelWindow<R, MODE == LM_REAL ? LM_DD : MODE>::LDS_FLOATS;
}
template <int R, int MODE>
__global__ void __launch_bounds__(64 * LEVEL_WAVES) __attribute__((amdgpu_waves_per_eu(R <= 4 ? 4 : 1))) k_level(const float* __restrict__ chns, float* __restrict__ pyr, float* __restrict__ rawOut,
    const LevelJob* __restrict__ jobs, const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it, const float* __restrict__ ft,
    int nChns, int64_t chns_fs, int64_t pyr_fs, float p, float* __restrict__ dump, int nFrames)
{
    const LevelJob J = jobs[blockIdx.y];
    constexpr int WF = levelWaveFloats<R, MODE>();
    __shared__ float ldsBlock[LEVEL_WAVES * WF];
    const int f = __builtin_amdgcn_readfirstlane(int(blockIdx.z) * LEVEL_WAVES + int(threadIdx.x >> 6));
    if (f >= nFrames)
    {
        return;
    }
    const LevelRank rk{};
    const LevelSeg sg{};
    level_body<R, MODE, LO_F32, 0>(J, f, chns, pyr, rawOut, descs, it, ft, nChns, chns_fs, pyr_fs, p, dump, ldsBlock, WF, rk, sg);
}

// All levels whose specialisation fits 128 VGPRs (R <= 4 in any mode, real levels up to R = 8) in ONE launch:
// blockIdx.z = level, longest plane chain first, blockIdx.y = group of LEVEL_WAVES frames, blockIdx.x = channel.  A plane
// is a sequential chain of wC column steps, so a launch lasts as long as its longest wave; as separate launches per
// (R, mode) on the 4 hardware queues the stage cost max-over-queues of a sum of such tails and kept 38 % of the wave slots
// busy (PMC SQ_WAVE_CYCLES, kernel trace).  In one grid the dispatcher starts the long chains first and back-fills slots
// with short ones as they free up.  The specialisation is picked by a workgroup-uniform switch.
// OUT & LO_RANK: every cell also (or only) leaves as its 16-bit threshold rank (host_plan.h) — the channel's tables sit in
// LDS behind the waves' rings (dynamic shared memory: LEVEL_WAVES * LEVEL_ALL_WF floats + maxRec records of 16 bytes).
#define ACF_LEVEL_KIND(R, M) ((R) * 8 + (M))
constexpr int LEVEL_ALL_WF = LevelWindow<4, LM_DD>::LDS_FLOATS; // the largest of the specialisations of k_level_all (R <= 4)
constexpr int LEVEL_ALL_WF_RANK = LevelWindow<4, LM_DD, LEVEL_RING_FLOATS_RANK>::LDS_FLOATS;
struct LevelRankArgs
{
    uint16_t* out;
    int64_t fs;
    const RankChan* chan;
    const RankRec* rec;
};
// SEG = 1: blockIdx.z = job * nSeg + segment; a job has min(nSeg, wC / (2 * warm)) segments (short levels stay one chain)
struct LevelSegArgs
{
    int32_t nSeg, warm, hMax, nJobs;
    float* spec;          // [frame][job][channel][segment][hMax]
    float* tru;
    int32_t* redo;        // repair launch (nSeg == 1): [frame][job][channel] != 0 -> recompute this plane; NULL: every plane
};
template <int OUT, int SEG>
__global__ void __launch_bounds__(64 * LEVEL_WAVES) __attribute__((amdgpu_waves_per_eu(4))) k_level_all(const float* __restrict__ chns, float* __restrict__ pyr, float* __restrict__ rawOut,
    const LevelJob* __restrict__ jobs, const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it, const float* __restrict__ ft,
    int nChns, int64_t chns_fs, int64_t pyr_fs, float p, float* __restrict__ dump, int nFrames, LevelRankArgs ra, LevelSegArgs sa)
{
    const int job = SEG ? int(blockIdx.z) / sa.nSeg : int(blockIdx.z);
    const LevelJob J = jobs[job];
    constexpr int RING = (OUT & LO_RANK) ? LEVEL_RING_FLOATS_RANK : LEVEL_RING_FLOATS;
    constexpr int WF = (OUT & LO_RANK) ? LEVEL_ALL_WF_RANK : LEVEL_ALL_WF;
    static_assert(LevelWindow<3, LM_DD, RING>::LDS_FLOATS <= WF && LevelWindow<4, LM_UU, RING>::LDS_FLOATS <= WF && LevelWindow<2, LM_DD, RING>::LDS_FLOATS <= WF &&
                      LevelWindow<3, LM_UU, RING>::LDS_FLOATS <= WF && LevelWindow<1, LM_DD, RING>::LDS_FLOATS <= WF,
        "ring size");
    extern __shared__ float ldsBlock[];
    LevelRank rk{};
    if ((OUT & LO_RANK) != 0)
    {
        uint4* recL = reinterpret_cast<uint4*>(ldsBlock + LEVEL_WAVES * WF);
        rk.fn = rank_tables_to_lds(ra.chan[blockIdx.x], ra.rec, recL, threadIdx.x, 64 * LEVEL_WAVES);
        rk.out = ra.out;
        rk.fs = ra.fs;
        rk.rec = recL;
        __syncthreads();
    }
    const int f = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * LEVEL_WAVES + int(threadIdx.x >> 6));
    if (f >= nFrames)
    {
        return;
    }
    LevelSeg sg{};
    if (SEG)
    {
        const int64_t plane = (int64_t(f) * sa.nJobs + job) * nChns + blockIdx.x;
        if (sa.redo)
        {
            // repair launch (one wave per plane): nothing to do when the plane's segments agreed; otherwise the flag is taken down
            // again for the next call (the flags are zero between calls: no launch clears them first)
            if (sa.redo[plane] == 0)
            {
                return;
            }
            if ((threadIdx.x & 63) == 0)
            {
                sa.redo[plane] = 0;
            }
        }
        const int seg = int(blockIdx.z) - job * sa.nSeg;
        const int nSegJ = max(1, min(sa.nSeg, J.wC / (2 * sa.warm)));
        if (seg >= nSegJ)
        {
            return;
        }
        const int segW = ((J.wC + nSegJ - 1) / nSegJ + 3) & ~3;
        sg.x0 = seg * segW;
        sg.x1 = min(sg.x0 + segW, J.wC);
        sg.xs = max(sg.x0 - sa.warm, 0);
        if (sg.x0 >= sg.x1)
        {
            return;
        }
        const bool lastSeg = sg.x1 >= J.wC;
        sg.spec = seg > 0 ? sa.spec + (plane * sa.nSeg + seg) * int64_t(sa.hMax) : nullptr;
        sg.tru = !lastSeg ? sa.tru + (plane * sa.nSeg + seg + 1) * int64_t(sa.hMax) : nullptr;
    }
#define LV_CASE(RR, MM)                                                                                                       \
    case ACF_LEVEL_KIND(RR, MM):                                                                                               \
        level_body<RR, MM, OUT, SEG>(J, f, chns, pyr, rawOut, descs, it, ft, nChns, chns_fs, pyr_fs, p, dump, ldsBlock, WF, rk, sg); \
        break;
#define LV_CASES(RR) LV_CASE(RR, LM_REAL) LV_CASE(RR, LM_DD) LV_CASE(RR, LM_UU)
    switch (J.kind)
    {
        LV_CASES(1)
        LV_CASES(2)
        LV_CASES(3)
        LV_CASES(4)
        LV_CASE(5, LM_REAL)
        LV_CASE(6, LM_REAL)
        LV_CASE(7, LM_REAL)
        LV_CASE(8, LM_REAL)
        default:
            break;
    }
#undef LV_CASES
#undef LV_CASE
}

// k_level_all's segments: spec (a segment's state after its warm-up) against tru (the previous segment's last column),
// bit for bit; the segment slots that were never written (a job with fewer segments) are skipped by the same rule the
// kernel used.  One workgroup per (job x segment, channel, frame).
__global__ void __launch_bounds__(64) k_level_verify(const float* __restrict__ spec, const float* __restrict__ tru, const LevelJob* __restrict__ jobs,
    LevelSegArgs sa, int nChns, int32_t* __restrict__ redo, int force)
{
    const int job = int(blockIdx.x) / sa.nSeg, seg = int(blockIdx.x) - job * sa.nSeg;
    const LevelJob J = jobs[job];
    const int nSegJ = max(1, min(sa.nSeg, J.wC / (2 * sa.warm)));
    const int segW = ((J.wC + nSegJ - 1) / nSegJ + 3) & ~3;
    if (seg == 0 || seg >= nSegJ || seg * segW >= J.wC)
    {
        return;
    }
    const int64_t plane = (int64_t(blockIdx.z) * sa.nJobs + job) * nChns + blockIdx.y;
    const uint32_t* a = reinterpret_cast<const uint32_t*>(spec) + (plane * sa.nSeg + seg) * int64_t(sa.hMax);
    const uint32_t* b = reinterpret_cast<const uint32_t*>(tru) + (plane * sa.nSeg + seg) * int64_t(sa.hMax);
    bool bad = force != 0;
    for (int y = threadIdx.x; y < J.hC; y += 64)
    {
        bad = bad || (a[y] != b[y]);
    }
    if (bad)
    {
        redo[plane] = 1;
    }
}

// imResample, exact 1/2 in both axes (imResampleMex.cpp:198-215, 284-288), ha % 4 == 0: 16 bytes per lane.
// A thread reads 4 source rows of the two source columns of its output column as float4 and writes 2 output
// rows as float2: out[y] = ((A[2x][2y] + A[2x+1][2y]) + (A[2x][2y+1] + A[2x+1][2y+1])) * rk — x pass then y pass,
// the association of the generic kernel's RS_EXACT path.
__global__ void __launch_bounds__(256) k_resample_half(const float* __restrict__ src, float* __restrict__ dst, const ResampleDesc* __restrict__ descs)
{
    const ResampleDesc& d = descs[0];
    const int ha = d.ha, hb = d.hb, wb = d.wb;
    const int hq = hb >> 1; // output row pairs
    const int64_t item = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t perPlane = int64_t(hq) * wb;
    if (item >= perPlane * d.nplanes)
    {
        return;
    }
    const int z = int(item / perPlane);
    const int rem = int(item - int64_t(z) * perPlane);
    const int xb = rem / hq, q = rem - xb * hq;
    const int ty = z < d.c1 ? 0 : (z < d.c2 ? 1 : 2);
    const float rk = d.rk[ty];
    const float* __restrict__ A = src + int64_t(blockIdx.z) * d.src_frame_stride + d.src_off + int64_t(z) * ha * d.wa + int64_t(2 * xb) * ha + 4 * q;
    const float4 p0 = *reinterpret_cast<const float4*>(A);
    const float4 p1 = *reinterpret_cast<const float4*>(A + ha);
    float2 o;
    o.x = ((p0.x + p1.x) + (p0.y + p1.y)) * rk;
    o.y = ((p0.z + p1.z) + (p0.w + p1.w)) * rk;
    float* __restrict__ B = dst + int64_t(blockIdx.z) * d.dst_frame_stride + d.dst_off + int64_t(z) * hb * wb + int64_t(xb) * hb + 2 * q;
    *reinterpret_cast<float2*>(B) = o;
}

// ------------------------------------------------------------------------
// LDCF decorrelation filters (BASELINE cfg 5; no reference counterpart — definition in include/acf_hip.h): one level's
// nChns planes convolved with k filters of 5x5 each, zero-padded 'same' true convolution:
//   out[f*nChns + c](y, x) = sum_{dx=-2..2} sum_{dy=-2..2} in[c](y - dy, x - dx) * filt[f][c][dx + 2][dy + 2]
// taps added in that order from 0 as one chain of f32 fused multiply-adds (acc = fmaf(v, w, acc)).  A thread produces one output cell; lanes run along image-y.
// ------------------------------------------------------------------------
struct LdcfJob
{
    int32_t h, w;        // padded level size (cells)
    int64_t inOff;       // level offset in one frame's pyramid
    int64_t outOff;      // level offset in one frame's filtered scratch (k * inOff)
};

// every level in one launch: blockIdx.z = frame * nLevels + level, blockIdx.y = output plane f*nChns + c
__global__ void __launch_bounds__(256) k_ldcf_conv(const float* __restrict__ pyr, float* __restrict__ out, const float* __restrict__ filt,
    const LdcfJob* __restrict__ jobs, int nLevels, int nChns, int64_t pyr_fs, int64_t out_fs)
{
    const int lvl = blockIdx.z % nLevels, frame = blockIdx.z / nLevels;
    const LdcfJob J = jobs[lvl];
    const int h = J.h, w = J.w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w)
    {
        return;
    }
    const int x = i / h, y = i - x * h;
    const int pc = blockIdx.y; // plane f*nChns + c
    const int c = pc % nChns;
    const float* __restrict__ in = pyr + int64_t(frame) * pyr_fs + J.inOff + int64_t(c) * h * w;
    const float* __restrict__ f = filt + int64_t(pc) * 25;
    float acc = 0.f;
#pragma unroll
    for (int dx = -2; dx <= 2; dx++)
    {
#pragma unroll
        for (int dy = -2; dy <= 2; dy++)
        {
            const int xx = x - dx, yy = y - dy;
            const bool ok = xx >= 0 && xx < w && yy >= 0 && yy < h;
            const float v = ok ? in[int64_t(min(max(xx, 0), w - 1)) * h + min(max(yy, 0), h - 1)] : 0.f;
            acc = __builtin_fmaf(v, f[(dx + 2) * 5 + (dy + 2)], acc); // (LDCF's taps are one chain of fused multiply-adds: this repo's definition)
        }
    }
    out[int64_t(frame) * out_fs + J.outOff + int64_t(pc) * h * w + i] = acc;
}

// ------------------------------------------------------------------------
// imResample on a source tile staged in LDS (k_ldcf_tile's halving of the filtered level; the image pyramid's down-sampled
// real scales take k_resample_strip below): x pass for every source row of the tile into a second LDS buffer, then the y pass.
// Arithmetic and association order are rs_C's and k_resample's (x pass then y pass), so results are bit-identical.
// ------------------------------------------------------------------------
constexpr int RT_YO = 64;

// A lane's y taps for the two passes on a source tile in LDS (T: [nCols][nRows], source rows rowLo.. / columns colLo..;
// C: [xo][nRows] x-pass buffer).
struct RtTaps
{
    int ya, q0, q1, ny;
    float wy[4];
    bool act, ySlow;
};

__device__ __forceinline__ RtTaps rt_taps(const ResampleDesc& d, const int32_t* __restrict__ it, const float* __restrict__ ft, int yb, int yb1, float r)
{
    RtTaps t;
    t.act = yb < yb1;
    const int ybc = t.act ? yb : yb1 - 1;
    t.q0 = t.q1 = 0;
    t.wy[0] = t.wy[1] = t.wy[2] = t.wy[3] = 0.f;
    t.ny = (d.ymode == RS_EXACT) ? d.yk : d.ybd0;
    t.ySlow = (d.ymode == RS_DOWN) && d.ybd0 > 4;
    if (d.ymode == RS_EXACT)
    {
        t.ya = d.yk * ybc;
    }
    else
    {
        t.q0 = it[d.y_start + ybc];
        t.q1 = it[d.y_start + ybc + 1];
        t.ya = it[d.y_src + t.q0];
        if (!t.ySlow)
        {
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                if (o < t.ny)
                {
                    t.wy[o] = ft[d.y_wt + t.q0 + o] * r; // ywts[y] *= r (:158-161)
                }
            }
        }
    }
    return t;
}

// The two passes on a source tile in LDS, for tiles of at most 16 output columns (k_ldcf_tile), on TWO planes at once: a cell of T / C is the
// pair {plane A, plane B} (k_ldcf_tile: two filters of one channel), every LDS access is 8 bytes and the arithmetic is packed f32 — per half
// exactly rs_C's / k_resample's products and sums (x pass then y pass, taps ascending; no fused multiply-add here).  Arranged for memory-level
// parallelism: a wave's four columns are in flight together and every tap read is unconditional (a tap beyond m / ny is read from wherever
// the index lands inside the workgroup's LDS and dropped by a select).  tP: row pitch of T and C in cells; BB == nullptr: plane B is not stored.
typedef float f2_t __attribute__((ext_vector_type(2)));
// a wave's KX output columns (wv + 4 k) of a tile: first source column (as an offset into T), tap count, weights — the same for every plane
// pair of the tile, so k_ldcf_tile reads them from the tile's records once
template <int KX>
struct RtCols
{
    int toff[KX], m[KX], wofs[KX];
    float w[KX][4];
};
template <int KX>
__device__ __forceinline__ RtCols<KX> rt_cols(const ResampleDesc& d, const int32_t* xrecTile, int nXo, int colLo, int tP)
{
    RtCols<KX> xc;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KX; k++)
    {
        const int32_t* rec = xrecTile + 8 * min(wv + 4 * k, nXo - 1);
        xc.toff[k] = (rec[0] - colLo) * tP;
        xc.m[k] = rec[1];
        xc.wofs[k] = rec[2];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            xc.w[k][j] = (d.xmode == RS_EXACT) ? 1.f : __int_as_float(rec[4 + j]);
        }
    }
    return xc;
}
template <int KX> // output columns per wave: tiles of up to 4 * KX output columns
__device__ __forceinline__ void rt_passes16x2(const ResampleDesc& d, const int32_t* __restrict__ it, const float* __restrict__ ft, const f2_t* T, f2_t* C,
    float* __restrict__ BA, float* __restrict__ BB, const RtTaps& tp, int yb, int xb0, int xb1, int rowLo, int nRows, int tP, float r, float rk, const RtCols<KX>& xc)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ha = d.ha, hb = d.hb;
    const int xmode = d.xmode, ymode = d.ymode;
    const int nXo = xb1 - xb0;
    const auto& toff = xc.toff;
    const auto& m = xc.m;
    const auto& wofs = xc.wofs;
    const auto& w = xc.w;
    for (int rr = lane; rr < nRows; rr += 64)
    {
        const bool below = rowLo + rr >= ha; // C[ha .. ha+3] = 0 (imResampleMex.cpp:133-137)
        f2_t t[KX][4];
#pragma unroll
        for (int k = 0; k < KX; k++)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                t[k][j] = T[toff[k] + j * tP + rr];
            }
        }
#pragma unroll
        for (int k = 0; k < KX; k++)
        {
            const int c = wv + 4 * k;
            f2_t s;
            if (xmode == RS_EXACT)
            {
                s = t[k][0] + t[k][1];
                const f2_t s2 = s + t[k][2];
                s = m[k] > 2 ? s2 : s;
                const f2_t s3 = s + t[k][3];
                s = m[k] > 3 ? s3 : s;
            }
            else
            {
                s = t[k][0] * w[k][0];
                const f2_t s1 = s + t[k][1] * w[k][1];
                s = m[k] > 1 ? s1 : s;
                const f2_t s2 = s + t[k][2] * w[k][2];
                s = m[k] > 2 ? s2 : s;
                const f2_t s3 = s + t[k][3] * w[k][3];
                s = m[k] > 3 ? s3 : s;
                for (int j = 4; j < m[k]; j++)
                {
                    s = s + T[toff[k] + j * tP + rr] * ft[wofs[k] + j];
                }
            }
            if (c < nXo)
            {
                C[c * tP + rr] = below ? f2_t{ 0.f, 0.f } : s;
            }
        }
    }
    __syncthreads();
    // y pass: lane = output row, the wave's four columns together
    if (!tp.act)
    {
        return;
    }
    const int ya = tp.ya, ny = tp.ny;
    if (tp.ySlow)
    {
        for (int c = wv; c < nXo; c += 4)
        {
            const f2_t* Cc = C + c * tP - rowLo;
            f2_t v = f2_t{ 0.f, 0.f };
            for (int q = tp.q0; q < tp.q1; q++)
            {
                v = v + Cc[it[d.y_src + q]] * (ft[d.y_wt + q] * r);
            }
            BA[int64_t(xb0 + c) * hb + yb] = v.x;
            if (BB)
            {
                BB[int64_t(xb0 + c) * hb + yb] = v.y;
            }
        }
        return;
    }
    f2_t u[KX][4];
#pragma unroll
    for (int k = 0; k < KX; k++)
    {
        const f2_t* Cc = C + min(wv + 4 * k, nXo - 1) * tP - rowLo;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            u[k][j] = Cc[ya + j];
        }
    }
#pragma unroll
    for (int k = 0; k < KX; k++)
    {
        const int c = wv + 4 * k;
        f2_t v;
        if (ymode == RS_EXACT)
        {
            f2_t sacc = u[k][0] + u[k][1];
            const f2_t s2 = sacc + u[k][2];
            sacc = ny > 2 ? s2 : sacc;
            const f2_t s3 = sacc + u[k][3];
            sacc = ny > 3 ? s3 : sacc;
            v = sacc * rk;
        }
        else
        {
            v = u[k][0] * tp.wy[0];
            v = v + u[k][1] * tp.wy[1];
            const f2_t v2 = v + u[k][2] * tp.wy[2];
            v = ny > 2 ? v2 : v;
            const f2_t v3 = v + u[k][3] * tp.wy[3];
            v = ny > 3 ? v3 : v;
        }
        if (c < nXo)
        {
            BA[int64_t(xb0 + c) * hb + yb] = v.x;
            if (BB)
            {
                BB[int64_t(xb0 + c) * hb + yb] = v.y;
            }
        }
    }
}


// ------------------------------------------------------------------------
// k_resample_strip: the down-sampling image resamples (chnsPyramid.cpp:310) as a march over STRIPS of output columns.
// Rounds 3-4 had a workgroup per 8 x 64-output tile (k_resample_tile, then a march over a row tile's column tiles with the next
// fill in flight, k_resample_march / _march2: deleted in round 5); it spent its time in a tile's dependent round trips (53 scalar
// loads, 135 branches, 1650 wave instructions per tile: 0.58 ms per 96 1080p frames for 8 MB per frame, 12 % of what the bytes
// need).  Here a workgroup takes a
// row tile of up to ~140 output rows (the whole source height of the tile: ~280 rows, one LDS column of `rowsP` floats) of one
// plane and walks its output columns RS_XO = 4 at a time (half as many of output B when the next real scale comes from the
// same source):
//   * the step's source columns arrive by LDS-DMA in 16-byte chunks, requested one step ahead into the other of two buffers;
//   * x pass: wave w = output column w of the step (its 32-byte column record one vector load, requested a step ahead), lanes =
//     source rows, every tap read issued before the first product; B's columns on wave pairs;
//   * y pass: (column, row) items dealt to the 256 threads once (a thread's rows — hence its taps — are the same in every step:
//     looked up before the march), one barrier between the passes, no table read inside the march.
// The products and sums are rs_C's and k_resample's (x pass then y pass, taps ascending), bit for bit.
// ------------------------------------------------------------------------
constexpr int RS_XO = 4;     // output columns of A per step (B: RS_XO / 2)
constexpr int RS_NT = 256;   // threads per workgroup
constexpr int RS_ITEMS = 3;  // y-pass items per thread at most ((RS_XO * yt + RS_XO / 2 * yt / 2) / RS_NT, rounded up)
constexpr int RS_KCH = 5;    // 64-row chunks of a source column at most (rowsP <= 320)
constexpr int RS_CP = 64 * (RS_KCH + 1);        // pitch of an x-pass column in LDS (every chunk's store, B's sixth included, lands in its own column)
constexpr int RS_REC = (RS_XO + RS_XO / 2) * 8; // ints of a step's column records
struct StripArgs
{
    const float* src;
    float* dstA;
    float* dstB;
    float* dump;            // >= RS_NT floats nobody reads
    const ResampleDesc* descs;
    const int32_t* it;
    const float* ft;
    int32_t descA, descB;   // descB < 0: one output
    int32_t yt, nty;        // A's output rows per row tile (even; B's tile: yt / 2 rows), row tiles
    int32_t nSteps, nSplit; // steps of RS_XO output columns of A; column segments per (plane, row tile)
    int32_t tileY;          // int arena: per row tile {rowLo (a multiple of 4), nRows (<= rowsP)}: the union of A's and B's source rows
    int32_t tileX;          // int arena: per step {colLo, nCols (<= maxCols)}
    int32_t rowsP, maxCols; // a source tile in LDS is [maxCols][rowsP] floats, rowsP % 4 == 0
    int32_t tileFloats;     // floats of one tile buffer: the requests of a tile, RS_NT chunks of 16 bytes per round, rounded up to whole rounds
    int32_t fillRounds;     // requests per wave and tile (a constant: the rounds past a tile's last chunk repeat it into the buffer's slack)
    int32_t ntyB, nStepsB;  // B's row tiles and steps (<= A's)
    int32_t slowRows;       // rows of the slow y pass's tap table in LDS (0: neither output takes it)
    uint32_t cpsMagic;      // ceil(2^32 / (rowsP / 4)): chunk index -> column by mulhi (checked on the host for every index)
};

// what one y-pass item needs in every step (looked up once): its rows of C, its taps
struct StripItem
{
    int32_t crow;     // float offset in C of the first tap: column * RS_CP + source row - rowLo
    int32_t ny;       // taps
    uint32_t offs;    // slow path: 4 bits per tap, tap j reads row crow + ((offs >> 4j) & 15) (fillers repeat a source row)
    int32_t col;      // column of the step: 0 .. RS_XO - 1 (A), RS_XO .. (B); -1: no item
    int32_t dst;      // yb (output row)
    int32_t slowRow;  // slow path: row of the tap table in LDS
    float w[4];       // the (at most four) weights; 1 for the exact form, whose sum is scaled by `gain` (r / k), else gain = 1
    float gain;
    uint32_t voff;    // byte offset of the item's output in its plane at step 0: (col * hb + yb) * 4 (the step adds a scalar)
};

// x pass of one output column for the NK row chunks k0, k0 + kStep, ... of the source tile T ([cols][rowsP], first column colLo)
// into the LDS column Cc (pitch RS_CP): rs_C's products and sums, straight-line — every tap read issued before the first
// product; a tap beyond m reads the last real tap's column and is dropped by a select; the exact form (sums of k columns,
// imResampleMex.cpp:198-215) is the weighted form with weights 1: t * 1.f == t for every t, so the sums are the same floats.
template <int NK>
__device__ __forceinline__ void strip_xpass(const float* T, float* Cc, bool exact, const int4& r0, const int4& r1, int colLo, int k0, int kStep, int rowsP,
    int rowLo, int ha)
{
    const int lane = threadIdx.x & 63;
    const int m = r0.y;
    const int toff = (r0.x - colLo) * rowsP;
    const float w0 = exact ? 1.f : __int_as_float(r1.x), w1 = exact ? 1.f : __int_as_float(r1.y), w2 = exact ? 1.f : __int_as_float(r1.z),
                w3 = exact ? 1.f : __int_as_float(r1.w);
    const int j1 = min(1, m - 1) * rowsP, j2 = min(2, m - 1) * rowsP, j3 = min(3, m - 1) * rowsP;
    float tv[NK][4];
#pragma unroll
    for (int k = 0; k < NK; k++)
    {
        const int rr = min(lane + 64 * (k0 + k * kStep), rowsP - 1);
        tv[k][0] = T[toff + rr];
        tv[k][1] = T[toff + j1 + rr];
        tv[k][2] = T[toff + j2 + rr];
        tv[k][3] = T[toff + j3 + rr];
    }
#pragma unroll
    for (int k = 0; k < NK; k++)
    {
        const int rr = lane + 64 * (k0 + k * kStep);
        float sv = tv[k][0] * w0;
        const float q1 = sv + tv[k][1] * w1;
        sv = m > 1 ? q1 : sv;
        const float q2 = sv + tv[k][2] * w2;
        sv = m > 2 ? q2 : sv;
        const float q3 = sv + tv[k][3] * w3;
        sv = m > 3 ? q3 : sv;
        Cc[rr] = (rowLo + rr >= ha) ? 0.f : sv; // C[ha .. ha+3] = 0 (imResampleMex.cpp:133-137); rows past the tile: never read
    }
}

template <bool HAVEB, bool SLOW>
__global__ void __launch_bounds__(RS_NT) k_resample_strip(StripArgs a)
{
    extern __shared__ float rs_lds[];
    const ResampleDesc& dA = a.descs[a.descA];
    const ResampleDesc& dB = a.descs[HAVEB ? a.descB : a.descA];
    const int32_t* __restrict__ it = a.it;
    const float* __restrict__ ft = a.ft;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int part = t % a.nSplit;
    t /= a.nSplit;
    const int ytile = t % a.nty;
    const int z = t / a.nty;
    const int perPart = (a.nSteps + a.nSplit - 1) / a.nSplit;
    const int s0 = part * perPart, s1 = min(s0 + perPart, a.nSteps);
    if (z >= dA.nplanes || s0 >= s1)
    {
        return;
    }
    // (every descriptor field the march needs, by value: a field selected between the two descriptors inside the loop would be
    // re-read from memory there — and an ordinary load's result used while LDS-DMA requests are in flight drains them all)
    const int ha = dA.ha, wa = dA.wa, hbA = dA.hb, wbA = dA.wb, hbB = dB.hb, wbB = dB.wb;
    const int xcolA = dA.x_col, xcolB = dB.x_col;
    const bool exA = dA.xmode == RS_EXACT, exB = dB.xmode == RS_EXACT;
    const int rowsP = a.rowsP, cps = rowsP >> 2;
    const int ty = z < dA.c1 ? 0 : (z < dA.c2 ? 1 : 2);
    const float rA = dA.r[ty], rkA = dA.rk[ty], rB = dB.r[ty], rkB = dB.rk[ty];
    const float* __restrict__ S = a.src + int64_t(blockIdx.z) * dA.src_frame_stride + dA.src_off + int64_t(z) * ha * wa;
    float* __restrict__ OA = a.dstA + int64_t(blockIdx.z) * dA.dst_frame_stride + dA.dst_off + int64_t(z) * hbA * wbA;
    float* __restrict__ OB = HAVEB ? a.dstB + int64_t(blockIdx.z) * dB.dst_frame_stride + dB.dst_off + int64_t(z) * hbB * wbB : a.dump;
    const srd_t Ssrd = make_srd(S, int64_t(ha) * wa * 4), Isrd = make_srd(it, int64_t(1) << 32);
    const srd_t OAsrd = make_srd(OA, int64_t(hbA) * wbA * 4), OBsrd = make_srd(OB, HAVEB ? int64_t(hbB) * wbB * 4 : 4);
    const int rowLo = it[a.tileY + 2 * ytile];
    const int ybA0 = ytile * a.yt, ytA = min(a.yt, hbA - ybA0);
    const bool tileB = HAVEB && ytile < a.ntyB;
    const int ybB0 = ytile * (a.yt >> 1), ytB = tileB ? min(a.yt >> 1, hbB - ybB0) : 0;
    const int nItA = RS_XO * ytA, nItB = (RS_XO / 2) * ytB;
    // LDS: two source tiles, the step's x-pass columns, two sets of column records, the slow y pass's taps, the step table
    float* const Tb0 = rs_lds;
    float* const C = rs_lds + 2 * size_t(a.tileFloats); // [RS_XO + RS_XO / 2][RS_CP]
    int32_t* const recL = reinterpret_cast<int32_t*>(C + (RS_XO + RS_XO / 2) * RS_CP); // [2][RS_REC]
    float* const slowW = reinterpret_cast<float*>(recL + 2 * RS_REC);                   // [max(slowRows, 1)][8]
    int32_t* const stepL = reinterpret_cast<int32_t*>(slowW + 8 * max(a.slowRows, 1));  // [steps of this segment + 2]{colLo, nCols}
    const bool slowA = SLOW && dA.ymode == RS_DOWN && dA.ybd0 > 4, slowB = SLOW && dB.ymode == RS_DOWN && dB.ybd0 > 4;

    // ---- this thread's y-pass items (a thread's rows are the same in every step)
    StripItem item[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++)
    {
        StripItem& q = item[k];
        const int i = tid + RS_NT * k;
        const bool isA = i < nItA, isB = !isA && i - nItA < nItB;
        q.col = -1;
        q.crow = 0;
        q.ny = 2;
        q.offs = 0;
        q.dst = 0;
        q.slowRow = 0;
        q.w[0] = q.w[1] = q.w[2] = q.w[3] = 0.f;
        q.gain = 1.f;
        if (isA || isB)
        {
            const ResampleDesc& d = isA ? dA : dB;
            const int yt_ = isA ? ytA : ytB, i_ = isA ? i : i - nItA;
            const int c = i_ / yt_, row = i_ - c * yt_;
            const int yb = (isA ? ybA0 : ybB0) + row;
            const float r = isA ? rA : rB;
            q.col = isA ? c : RS_XO + c;
            q.dst = yb;
            if (d.ymode == RS_EXACT)
            {
                q.ny = d.yk;
                q.crow = d.yk * yb - rowLo;
                q.w[0] = q.w[1] = q.w[2] = q.w[3] = 1.f; // (c * 1.f == c: the sums of imResampleMex.cpp:286,:309,:316)
                q.gain = isA ? rkA : rkB;
            }
            else
            {
                const int q0 = it[d.y_start + yb], q1 = it[d.y_start + yb + 1];
                const int ya = it[d.y_src + q0];
                q.crow = ya - rowLo;
                if (SLOW && d.ybd0 > 4)
                {
                    // more than four taps (the reference's generic loop, imResampleMex.cpp:357-370): weights in the LDS table, one row
                    // per output row of the tile (A's rows first), written by the item of the row's first column
                    q.ny = q1 - q0; // (<= 8, each within 15 rows of the first: the host planned it)
                    q.slowRow = (isA ? 0 : (slowA ? ytA : 0)) + row;
#pragma unroll
                    for (int j = 0; j < 8; j++)
                    {
                        float wj = 0.f;
                        if (j < q1 - q0)
                        {
                            wj = ft[d.y_wt + q0 + j] * r;
                            q.offs |= uint32_t(it[d.y_src + q0 + j] - ya) << (4 * j);
                        }
                        if (c == 0)
                        {
                            slowW[q.slowRow * 8 + j] = wj;
                        }
                    }
                }
                else
                {
                    q.ny = d.ybd0;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        if (j < d.ybd0)
                        {
                            q.w[j] = ft[d.y_wt + q0 + j] * r; // ywts[y] *= r (:158-161)
                        }
                    }
                }
            }
            q.crow += q.col * RS_CP;
            q.voff = uint32_t(c * (isA ? hbA : hbB) + yb) * 4u;
        }
    }
    // which of the thread's item slots hold an item of the slow form anywhere in this WAVE (wave-uniform: the slow form's extra
    // reads are skipped where no lane needs them — A's items fill the first slots, B's the last)
    bool slowK[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++)
    {
        const bool mine = SLOW && item[k].col >= 0 && (item[k].col >= RS_XO ? slowB : slowA);
        slowK[k] = __builtin_amdgcn_ballot_w64(mine) != 0;
    }
    // fill: a lane's chunk of a round is the same (column of the tile, rows) in every step
    uint32_t fCol[4], fRow[4]; // (a.fillRounds <= 4: the host planned it)
#pragma unroll
    for (int rnd = 0; rnd < 4; rnd++)
    {
        const uint32_t q = uint32_t(rnd * RS_NT + wv * 64 + lane);
        const uint32_t col = __umulhi(q, a.cpsMagic);
        const uint32_t ch = q - col * uint32_t(cps);
        fCol[rnd] = col;
        fRow[rnd] = uint32_t(min(rowLo + 4 * int(ch), ha - 4)) * 4u; // rows past the image: clamped duplicates (the x pass zeroes them)
    }
    const uint32_t colBytes = uint32_t(ha) * 4u;
    // the segment's step table {colLo, nCols} into LDS (two entries past the end repeat the last step)
    for (int i = tid; i < 2 * (s1 - s0 + 2); i += RS_NT)
    {
        stepL[i] = it[a.tileX + 2 * min(s0 + (i >> 1), a.nSteps - 1) + (i & 1)];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // (the step table and the slow taps are every wave's)

    // ---- the march.  No ordinary load, no branch inside: a tile's requests are a fixed number per wave (a.fillRounds, + one for
    // wave 0's column records), its stores RS_ITEMS per wave.
    auto fill = [&](int s, float* T) {
        const float* stepF = reinterpret_cast<const float*>(stepL); // (as floats: see the column records below)
        const int colLo = __builtin_amdgcn_readfirstlane(__float_as_int(stepF[2 * (s - s0)])),
                  nCols = __builtin_amdgcn_readfirstlane(__float_as_int(stepF[2 * (s - s0) + 1]));
#pragma unroll
        for (int rnd = 0; rnd < 4; rnd++)
        {
            if (rnd < a.fillRounds) // (uniform)
            {
                // (a chunk past the tile's last column: the last column's rows again, into the buffer's slack)
                const uint32_t x = uint32_t(min(colLo + int(min(fCol[rnd], uint32_t(nCols - 1))), wa - 1));
                // (the BUFFER form: after a global_load_lds the compiler drains every request before the kernel's next LDS access — it
                // cannot tell the tile buffers apart — and the look-ahead is gone; k_level's ring has the same reason)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(Ssrd, (lptr_t)(T + 4u * uint32_t(rnd * RS_NT + wv * 64)), 16, x * colBytes + fRow[rnd], 0, 0, 0);
            }
        }
        // the step's column records — A's columns RS_XO s .., then B's (clamped to the last column: computed, not stored) — one dword per
        // lane of wave 0
        if (wv == 0 && lane < RS_REC) // (the request writes LDS at dst + 4 * lane: the lanes past the records must not take part)
        {
            const int e = lane, cIdx = e >> 3, wIdx = e & 7;
            const bool forB = cIdx >= RS_XO;
            const int x = forB ? min((RS_XO / 2) * s + cIdx - RS_XO, wbB - 1) : min(RS_XO * s + cIdx, wbA - 1);
            int32_t* dst = recL + ((s - s0) & 1) * RS_REC;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Isrd, (lptr_t)dst, 4, uint32_t((forB ? xcolB : xcolA) + 8 * x + wIdx) * 4u, 0, 0, 0);
        }
        return colLo;
    };
    int colLoNext = fill(s0, Tb0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int s = s0; s < s1; s++)
    {
        // Tile s has arrived: this wave's requests for it are older than its RS_ITEMS stores of the previous step's y pass and
        // requests complete in order, so the stores may stay in flight; the barrier makes the arrival every wave's.  Every wave is
        // also past the previous step's y pass: C, the other tile buffer and the other record set may be written again.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RS_ITEMS * (HAVEB ? 2 : 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        const int cur = (s - s0) & 1;
        const int colLoCur = colLoNext;
        // the next tile (past the last step: the table's repeat of it, into the buffer nobody reads any more)
        colLoNext = fill(s + 1, Tb0 + (cur ^ 1) * a.tileFloats);
        const float* T = Tb0 + cur * a.tileFloats;
        // (read as FLOATS: the compiler orders an LDS read behind every LDS-DMA request in flight unless type-based alias analysis
        // separates the two, and the requests are typed as ints — an int4 read here costs an s_waitcnt vmcnt(0), i.e. the look-ahead)
        const float* rl = reinterpret_cast<const float*>(recL) + cur * RS_REC;
        auto rec4 = [&](int i) { return make_int4(__float_as_int(rl[4 * i]), __float_as_int(rl[4 * i + 1]), __float_as_int(rl[4 * i + 2]), __float_as_int(rl[4 * i + 3])); };
        // x pass: A's column wv; B's column wv >> 1, the even / odd row chunks on the two waves of a pair
        {
            const int4 r0 = rec4(2 * wv), r1 = rec4(2 * wv + 1);
            strip_xpass<RS_KCH>(T, C + wv * RS_CP, exA, r0, r1, colLoCur, 0, 1, rowsP, rowLo, ha);
        }
        const bool stepB = HAVEB && tileB && s < a.nStepsB;
        if (HAVEB)
        {
            const int4 r0 = rec4(2 * (RS_XO + (wv >> 1))), r1 = rec4(2 * (RS_XO + (wv >> 1)) + 1);
            strip_xpass<(RS_KCH + 1) / 2>(T, C + (RS_XO + (wv >> 1)) * RS_CP, exB, r0, r1, colLoCur, wv & 1, 2, rowsP, rowLo, ha);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // y pass: exactly RS_ITEMS stores per output, wave and step
        const int xbA = RS_XO * s, xbB = (RS_XO / 2) * s;
        const uint32_t stepA = uint32_t(xbA * hbA) * 4u, stepB4 = uint32_t(xbB * hbB) * 4u; // (scalar offsets of the step's first columns)
#pragma unroll
        for (int k = 0; k < RS_ITEMS; k++)
        {
            const StripItem& q = item[k];
            const bool isB = q.col >= RS_XO;
            const int cA = xbA + q.col, cB = xbB + q.col - RS_XO;
            const bool on = q.col >= 0 && (isB ? (stepB && cB < wbB) : cA < wbA);
            const float* Cc = C + q.crow;
            const int ny = q.ny;
            const float c0 = Cc[0], c1 = Cc[1], c2 = Cc[min(2, ny - 1)], c3 = Cc[min(3, ny - 1)];
            float v = c0 * q.w[0];
            v = v + c1 * q.w[1];
            const float v2 = v + c2 * q.w[2];
            v = ny > 2 ? v2 : v;
            const float v3 = v + c3 * q.w[3];
            v = ny > 3 ? v3 : v;
            v = v * q.gain; // (1.f unless the exact form: v * 1.f == v)
            if (SLOW && slowK[k]) // (wave-uniform)
            {
                // both forms for every lane of such a wave, one select (a wave's items may be of both outputs: no divergent branch)
                float c[8];
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    c[j] = Cc[(q.offs >> (4 * j)) & 15u];
                }
                const float* wr = slowW + q.slowRow * 8; // (scalar float reads: a float4 read is ordered behind the requests in flight, see above)
                const float w8[8] = { wr[0], wr[1], wr[2], wr[3], wr[4], wr[5], wr[6], wr[7] };
                float vs = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    const float vj = vs + c[j] * w8[j]; // from 0.f, taps ascending (imResampleMex.cpp:357-370)
                    vs = j < ny ? vj : vs;
                }
                v = (isB ? slowB : slowA) ? vs : v;
            }
            // BUFFER stores, one per output: requests of one kind complete in order, which the counted wait at the top of the step
            // relies on (a global_store between buffer loads may overtake them); a lane without an item of that output stores
            // beyond the descriptor's range, i.e. nowhere
            buf_st(OAsrd, (on && !isB) ? q.voff : 0xfffffff0u, stepA, v);
            if (HAVEB)
            {
                buf_st(OBsrd, (on && isB) ? q.voff : 0xfffffff0u, stepB4, v);
            }
        }
    }
}

// ------------------------------------------------------------------------
// LDCF (BASELINE cfg 5: "k 5x5 per-channel decorrelation filters fused into the pyramid kernel"): one workgroup turns a
// tile of ONE channel plane of one pyramid level into the k filtered AND halved planes of the LDCF pyramid —
//   C_f = conv2(plane, filter_f, 'same')  (zero padded, taps in k_ldcf_conv's order: dx then dy ascending)
//   out_f = imResample(C_f, .5)           (the x pass / y pass on a tile in LDS: rt_passes16x2)
// The plane tile (+2 cells of halo, zeros outside the plane) is read once into LDS, each filter's conv result is
// written to the LDS source tile of the resample and never reaches HBM: the separate k_ldcf_conv + k_resample pair
// wrote and re-read k full-resolution copies of the pyramid (760 MB per 4K frame at k = 4).  A job is one output tile of
// one level (flat list built at plan time: no empty blocks); blockIdx.y = input channel, blockIdx.z = frame.
// The filters of a channel are taken TWO AT A TIME, as the halves of packed f32 (round 5; rounds 3-4 packed two tile rows of one
// filter): both filters read the same plane cells, so a cell is read from LDS once for the pair, a thread's cells arrive as
// 8-byte row pairs, and the pair's results travel through the resample's passes as {filter A, filter B} cells.
// ------------------------------------------------------------------------
struct LdcfTileJob
{
    int32_t level;          // level = LDCF descriptor index
    int32_t ytile, xtile;   // output tile (yo rows x xo columns)
    int32_t tile_y, tile_x; // int-arena offsets of the level's {rowLo,rowHi} / {colLo,colHi} tables
    int32_t yo, xo;         // output rows / columns per tile of this level (<= 64 x 16, source tile <= 128 rows x 32 columns)
    int32_t pad_;
};

// floats of LDS k_ldcf_tile needs for source tiles of at most maxRows x maxCols cells and xo output columns
__host__ __device__ inline size_t ldcfTileLdsFloats(int maxRows, int maxCols, int xo)
{
    const size_t tP = size_t(maxRows + 1) & ~size_t(1);
    return 2 * size_t(maxCols) * tP + 2 * size_t(xo) * tP + 8 * size_t(xo) + size_t(maxCols + 4) * (size_t(maxRows + 6) & ~size_t(1)) + 64;
}

template <int LN, int KX> // LN: output columns per thread of the filter stage (LN + 4 tile columns are read for them); KX: rt_passes16x2's
__global__ void __launch_bounds__(256) k_ldcf_tile(const float* __restrict__ pyr, float* __restrict__ out, const float* __restrict__ filt,
    const LdcfTileJob* __restrict__ jobs, const LdcfJob* __restrict__ levels, const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it,
    const float* __restrict__ ft, int maxRows, int maxCols, int xo, int K, int nChns, int64_t pyr_fs)
{
    extern __shared__ __attribute__((aligned(16))) float lt_lds[];
    const LdcfTileJob J = jobs[blockIdx.x];
    const ResampleDesc& d = descs[J.level];
    const LdcfJob L = levels[J.level];
    const int c = blockIdx.y;
    const int ha = d.ha, hb = d.hb, wa = d.wa, wb = d.wb;
    const int lane = threadIdx.x & 63;
    const int yb0 = J.ytile * J.yo, yb1 = min(yb0 + J.yo, hb);
    const int xb0 = J.xtile * J.xo, xb1 = min(xb0 + J.xo, wb);
    const float r = d.r[0], rk = d.rk[0];
    const int rowLo = it[J.tile_y + 2 * J.ytile], rowHi = it[J.tile_y + 2 * J.ytile + 1];
    const int colLo = it[J.tile_x + 2 * J.xtile], colHi = it[J.tile_x + 2 * J.xtile + 1];
    const int nRows = min(rowHi - rowLo + 1, maxRows), nCols = min(colHi - colLo + 1, maxCols);
    const int tP = (maxRows + 1) & ~1;                                        // row pitch of T and C (cells; even: a row pair is 16 aligned bytes)
    f2_t* T = reinterpret_cast<f2_t*>(lt_lds);                                // [nCols][tP] filtered tile {filter A, filter B} = source tile of the resample
    f2_t* C = T + size_t(maxCols) * tP;                                       // [xo][tP] x-pass columns
    int32_t* recL = reinterpret_cast<int32_t*>(C + size_t(xo) * tP);          // [xo][8] x-pass column records of this tile
    float* P = reinterpret_cast<float*>(recL + 8 * xo);                       // [nCols + 4][pR] plane tile with halo (+ 64 floats of slack behind it)
    const int pR = (nRows + 6) & ~1;                                          // >= nRows + 5 (the second row of the last pair reads one row further), even
    if (int(threadIdx.x) < 8 * (xb1 - xb0))
    {
        recL[threadIdx.x] = it[d.x_col + 8 * xb0 + threadIdx.x]; // read once: every filter's x pass uses them
    }
    const int yb = yb0 + lane;
    const RtTaps tp = rt_taps(d, it, ft, yb, yb1, r);
    const float* __restrict__ A = pyr + int64_t(blockIdx.z) * pyr_fs + L.inOff + int64_t(c) * ha * wa;
    // LDS-DMA, 64 consecutive tile cells per wave instruction, the whole tile in flight at once (a load -> ds_write loop exposed one
    // memory round trip per 256 cells).  Buffer form: a cell outside the plane gets an offset beyond the descriptor's range and
    // arrives as 0 — no LDS writes between the requests (each one made the compiler drain the requests before it: DESIGN 3.0)
#ifndef ACF_LDCF_NO_FILL
    {
        // a wave instruction = 64 rows of ONE padded column: the column's byte offset is scalar, a lane's row offset one of three
        // values — no per-cell index arithmetic (the flat form spent 42 instructions per request on i / pR)
        const srd_t Asrd = make_srd(A, int64_t(ha) * wa * 4);
        const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
        uint32_t yoff[3];
#pragma unroll
        for (int part = 0; part < 3; part++)
        {
            const int y = rowLo + 64 * part + lane - 2;
            yoff[part] = (y >= 0 && y < ha) ? uint32_t(y) * 4u : 0xfffffff0u;
        }
        for (int cc = wv; cc < nCols + 4; cc += 4)
        {
            const int x = colLo + cc - 2;
            const bool xok = x >= 0 && x < wa;
            const uint32_t soff = xok ? uint32_t(x) * uint32_t(ha) * 4u : 0u;
            float* Pc = P + cc * pR;
#pragma unroll
            for (int part = 0; part < 3; part++)
            {
                if (64 * part + lane < pR) // (the last part's lanes beyond the column write nothing: the next column starts there)
                {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(Asrd, (lptr_t)(Pc + 64 * part), 4, xok ? yoff[part] : 0xfffffff0u, soff, 0, 0);
                }
            }
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int RPN = (nRows + 1) >> 1; // row pairs of the tile (<= 64)
    const RtCols<KX> xc = rt_cols<KX>(d, recL, xb1 - xb0, colLo, tP);
    for (int f = 0; f < K; f += 2)
    {
        const bool haveB = f + 1 < K; // (an odd k: the last pair's second filter has zero taps and is not stored)
        const int pcA = f * nChns + c, pcB = (haveB ? f + 1 : f) * nChns + c;
        // 2 x 25 taps through the scalar unit (wave-uniform addresses), kept as 25 VGPR pairs {filter A's tap, filter B's tap}
        typedef const __attribute__((address_space(4))) float* cfp_t;
        cfp_t fwA = (cfp_t)(uintptr_t)(filt + int64_t(pcA) * 25), fwB = (cfp_t)(uintptr_t)(filt + int64_t(pcB) * 25);
        f2_t wp[25];
#pragma unroll
        for (int k = 0; k < 25; k++)
        {
            wp[k] = f2_t{ fwA[k], haveB ? fwB[k] : 0.f };
            asm volatile("" : "+v"(wp[k])); // (in VGPRs: a VALU instruction with an SGPR operand issues at 1.7x the cost of one without)
        }
        // A thread takes LN consecutive COLUMNS at the tile's row pair {2 lp, 2 lp + 1} (lanes along the rows: conflict-free 8-byte LDS
        // reads; the plan keeps a tile within 128 rows x 32 columns, so the workgroup's 256 items are the whole tile at LN = 8).  A column's
        // six cells (padded rows 2 lp .. 2 lp + 5) arrive as three aligned pairs; a tap is one v_pk_fma_f32 per output cell: the cell's half
        // of its pair broadcast to both halves (op_sel), times {A's tap, B's tap}.  Per output the taps are still added dx then dy ascending,
        // as ONE chain of fused multiply-adds starting from 0 (k_ldcf_conv's order; each half of v_pk_fma_f32 = C's fmaf): columns are
        // therefore consumed from cc + LN + 3 down to cc — output j meets column q at dx = j + 2 - q —, each read once, two columns ahead
        // of its use (the scheduling barriers keep three columns live), and dropped.
#define LDCF_PKFMA(ACC, V, HALF, K)                                                                                                      \
    if ((HALF) & 1)                                                                                                                      \
    {                                                                                                                                    \
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(ACC) : "v"(V), "v"(wp[K]));                          \
    }                                                                                                                                    \
    else                                                                                                                                 \
    {                                                                                                                                    \
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(ACC) : "v"(V), "v"(wp[K]));                          \
    }
#define LDCF_LOAD(Q)                                                                                                 \
    {                                                                                                                \
        const f2_t* pc0 = reinterpret_cast<const f2_t*>(P + min(cc + (Q), nCols + 3) * pR + 2 * lp);                 \
        v[Q][0] = pc0[0];                                                                                            \
        v[Q][1] = pc0[1];                                                                                            \
        v[Q][2] = pc0[2];                                                                                            \
    }
#ifdef ACF_LDCF_NO_CONV
        const int nQ = 0;
#else
        const int nQ = (nCols + LN - 1) / LN;
#endif
        for (int i = threadIdx.x; i < nQ * 64; i += 256)
        {
            const int cq = i >> 6, lp = i & 63, cc = cq * LN;
            if (lp >= RPN)
            {
                continue;
            }
            f2_t acc[LN][2]; // [output column][row of the pair] = {filter A, filter B}
#pragma unroll
            for (int j = 0; j < LN; j++)
            {
                acc[j][0] = acc[j][1] = f2_t{ 0.f, 0.f };
            }
            f2_t v[LN + 4][3]; // [column cc + q of the padded tile (clamped past its end: never stored)][padded rows 2 lp + {0 1, 2 3, 4 5}]
            LDCF_LOAD(LN + 3);
            LDCF_LOAD(LN + 2);
#pragma unroll
            for (int q = LN + 3; q >= 0; q--)
            {
                if (q >= 2)
                {
                    LDCF_LOAD(q - 2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < LN; j++)
                {
                    const int dx = j + 2 - q;
                    if (dx < -2 || dx > 2)
                    {
                        continue;
                    }
#pragma unroll
                    for (int dy = -2; dy <= 2; dy++)
                    {
                        // tile row y meets padded row y + 2 - dy: the pair's first row reads cell 2 - dy of the six, its second 3 - dy
                        LDCF_PKFMA(acc[j][0], v[q][(2 - dy) >> 1], (2 - dy) & 1, (dx + 2) * 5 + (dy + 2));
                        LDCF_PKFMA(acc[j][1], v[q][(3 - dy) >> 1], (3 - dy) & 1, (dx + 2) * 5 + (dy + 2));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < LN; j++)
            {
                if (cc + j < nCols)
                {
                    // (a last odd row's partner is written too — inside the pitch, never read)
                    typedef float f4_t __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<f4_t*>(T + (cc + j) * tP + 2 * lp) = f4_t{ acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y };
                }
            }
        }
#undef LDCF_LOAD
#undef LDCF_PKFMA
        __syncthreads();
        float* __restrict__ B0 = out + int64_t(blockIdx.z) * d.dst_frame_stride + d.dst_off;
        // (no barrier after the y pass: it reads C only, the next pair's conv writes T only, and the x pass that rewrites C
        // comes after the barrier that follows that conv)
#ifndef ACF_LDCF_NO_PASSES
        rt_passes16x2<KX>(d, it, ft, T, C, B0 + int64_t(pcA) * hb * wb, haveB ? B0 + int64_t(pcB) * hb * wb : nullptr, tp, yb, xb0, xb1, rowLo, nRows, tP, r, rk, xc);
#endif
    }
}

// grid.x needed for one descriptor
static inline int resampleBlocks(const ResampleDesc& d, int xt = RS_XT)
{
    return ((d.hb + 63) / 64) * ((d.wb + 4 * xt - 1) / (4 * xt)) * d.nplanes;
}

// ------------------------------------------------------------------------
// The cascade: ParallelDetectionBody::operator()/evaluate
// (toolbox/acfDetect1.cpp:84-138) for every window of every level of every
// frame.  One lane per window; in the first stage lanes are consecutive along
// r (the contiguous image-y axis), so each feature fetch of a wave is one
// contiguous segment of the level's channel buffer.
//
// Staging.  A window's score is a running sum that stops at the first
// h <= cascThr; on the headline workload 76 % of the windows are gone after
// 16 trees and 99.5 % after 64, but a wave lives as long as its longest lane
// (131 trees on average).  The tree range is therefore cut into stages
// [0,16) [16,32) [32,128) [128,nTrees): after each stage the surviving lanes
// are compacted (wave ballot + prefix count, one atomic per wave) into a
// per-frame queue of {level, window, h}, and the next stage runs dense waves
// over that queue.  Scores are unaffected: each window still adds the same
// leaves in the same order.
//
// Depth-2 fast path.  A node's feature id is kept as packed (z, c, r); its
// channel offset z*area + c*hP + r is rebuilt from the lane's level geometry,
// so ONE level-independent node table serves every level and every stage, and
// tree t's three nodes / four leaves are wave-uniform scalar loads.  Feature
// addresses do not depend on h, so the loads of CG consecutive trees (three
// per tree: root and both children) are issued together before the
// comparisons are resolved in order — the dependent-load chain per tree
// becomes CG*3 independent loads in flight.
// ------------------------------------------------------------------------
struct CascLevel
{
    int32_t hP, wP, nWinR, nWinC;
    int32_t firstBlock; // first block index of this level inside one frame's stage-0 grid
    int32_t nWin;
    int64_t off;        // level offset in the fused pyramid
    int64_t nodeOff;    // generic path: offset of this level's cid table
    int64_t offR;       // rank pyramid (16-bit cells): cell offset of the level inside one frame (a multiple of 8)
    int32_t pitchR;     // rank pyramid: cells between columns (hP rounded up to 8: every column starts on 16 bytes)
    int32_t padR_;
};

struct __attribute__((aligned(16))) CascNode2
{
    uint32_t zcr[4]; // (z << 24) | (c << 12) | r for nodes 0,1,2; [3] unused
    float thr[4];    // thr[3] unused
    float hs[4];     // leaves 3..6
};

struct CascArgs
{
    const float* pyr;
    int64_t pyr_fs;
    const CascLevel* levels;
    const int32_t* blockLevel; // stage-0 block -> level
    int32_t blocksPerFrame, nFrames;
    int32_t nTrees, nTreeNodes, treeDepth;
    int32_t stride, shrink;
    int32_t mH, mW, nChns;   // model window in cells (modelDsPad / shrink), channels
    float cascThr;
    // generic path tables
    const uint32_t* cidAll;  // [level][nTrees*nTreeNodes]
    const uint32_t* fids;    // [nTrees*nTreeNodes] raw feature ids (tail stage)
    const float* thrs;       // [nTrees*nTreeNodes]
    const float* hs;
    const uint32_t* child;
    const CascNode2* nodes2; // depth-2 packed table [nTrees]
    // stage
    int32_t t0, t1;          // tree range of this stage
    int32_t last;            // t1 == nTrees: survivors are hits
    const uint2* qin;        // [frame][qcap] {(level << 24) | window, h bits}
    const int32_t* qinCount; // [frame]
    uint2* qout;
    int32_t* qoutCount;
    int32_t qcap;
    // output
    acf_hip_hit* hits; // [frame][maxHits]
    int32_t* counts;   // [frame]
    int32_t maxHits;
    // last stage of a fixed-depth model as leaf codes + ordered scan (k_tail_codesD / k_tail_scanD): the first codeCap
    // queue entries of a frame; k_cascade_tail then starts at entry qskip
    uint8_t* codes;    // [frame][codeCap][codePitch]: 4 * (leaf index) of tree t0 + j of entry i
    int32_t codeCap, codePitch, qskip;
};

#define CASC_CG 4

template <int MODE> // 2: packed depth-2 path; 1: generic fixed depth; 0: child walk
__device__ __forceinline__ void casc_eval(const CascArgs& a, const float* __restrict__ chn, int hP, int area, int64_t nodeOff, float& h, bool& alive)
{
    const float thrC = a.cascThr;
    if (MODE == 2)
    {
        const CascNode2* __restrict__ nodes = a.nodes2;
        int t = a.t0;
        for (; t + CASC_CG <= a.t1; t += CASC_CG)
        {
            if (!__any(alive))
            {
                return;
            }
            CascNode2 nd[CASC_CG];
#pragma unroll
            for (int g = 0; g < CASC_CG; g++)
            {
                nd[g] = nodes[t + g]; // uniform address: scalar loads
            }
            // Issue all CG*3 feature loads first (addresses do not depend on h), then
            // resolve the trees in order with selects only: one basic block, so the
            // loads stay batched instead of being sunk behind per-tree branches.
            float f0[CASC_CG], f1[CASC_CG], f2[CASC_CG];
#pragma unroll
            for (int g = 0; g < CASC_CG; g++)
            {
                f0[g] = f1[g] = f2[g] = 0.f;
            }
            if (alive)
            {
#pragma unroll
                for (int g = 0; g < CASC_CG; g++)
                {
                    const uint32_t a0 = nd[g].zcr[0], a1 = nd[g].zcr[1], a2 = nd[g].zcr[2];
                    f0[g] = chn[(a0 >> 24) * area + ((a0 >> 12) & 0xfff) * hP + (a0 & 0xfff)];
                    f1[g] = chn[(a1 >> 24) * area + ((a1 >> 12) & 0xfff) * hP + (a1 & 0xfff)];
                    f2[g] = chn[(a2 >> 24) * area + ((a2 >> 12) & 0xfff) * hP + (a2 & 0xfff)];
                }
            }
#pragma unroll
            for (int g = 0; g < CASC_CG; g++)
            {
                const bool lt0 = f0[g] < nd[g].thr[0];
                const float fc = lt0 ? f1[g] : f2[g];
                const float th1 = lt0 ? nd[g].thr[1] : nd[g].thr[2];
                const bool lt1 = fc < th1;
                // k after two steps: lt0 ? (lt1 ? 3 : 4) : (lt1 ? 5 : 6)
                const float hv = lt0 ? (lt1 ? nd[g].hs[0] : nd[g].hs[1]) : (lt1 ? nd[g].hs[2] : nd[g].hs[3]);
                const float hn = h + hv;
                h = alive ? hn : h;          // a rejected window keeps the score it was rejected with
                alive = alive && (hn > thrC);
            }
        }
        for (; t < a.t1; t++)
        {
            if (!__any(alive))
            {
                return;
            }
            const CascNode2 n1 = nodes[t];
            if (alive)
            {
                const uint32_t a0 = n1.zcr[0], a1 = n1.zcr[1], a2 = n1.zcr[2];
                const float g0 = chn[(a0 >> 24) * area + ((a0 >> 12) & 0xfff) * hP + (a0 & 0xfff)];
                const bool lt0 = g0 < n1.thr[0];
                const uint32_t ac = lt0 ? a1 : a2;
                const float fc = chn[(ac >> 24) * area + ((ac >> 12) & 0xfff) * hP + (ac & 0xfff)];
                const float th1 = lt0 ? n1.thr[1] : n1.thr[2];
                const bool lt1 = fc < th1;
                const float hv = lt0 ? (lt1 ? n1.hs[0] : n1.hs[1]) : (lt1 ? n1.hs[2] : n1.hs[3]);
                h += hv;
                alive = h > thrC;
            }
        }
    }
    else if (MODE == 1)
    {
        const uint32_t* cid = a.cidAll + nodeOff;
        const int D = a.treeDepth;
        for (int t = a.t0; t < a.t1; t++)
        {
            if (!__any(alive))
            {
                return;
            }
            if (alive)
            {
                const uint32_t offset = uint32_t(t) * uint32_t(a.nTreeNodes);
                uint32_t k = offset, k0 = 0;
                for (int i = 0; i < D; i++)
                {
                    const float ftr = chn[cid[k]];
                    k = (ftr < a.thrs[k]) ? 1 : 2;
                    k0 = k += k0 * 2;
                    k += offset;
                }
                h += a.hs[k];
                alive = h > thrC;
            }
        }
    }
    else
    {
        const uint32_t* cid = a.cidAll + nodeOff;
        for (int t = a.t0; t < a.t1; t++)
        {
            if (!__any(alive))
            {
                return;
            }
            if (alive)
            {
                const uint32_t offset = uint32_t(t) * uint32_t(a.nTreeNodes);
                uint32_t k = offset, k0 = offset;
                while (a.child[k])
                {
                    const float ftr = chn[cid[k]];
                    k = (ftr < a.thrs[k]) ? 1 : 0;
                    k0 = k = a.child[k0] - k + offset;
                }
                h += a.hs[k];
                alive = h > thrC;
            }
        }
    }
}

// Survivors of a stage: hits if this was the last stage, else queue entries.
// Compaction is two-level: a ballot prefix inside each wave, the four wave
// totals combined through LDS, ONE atomic per workgroup on the frame's counter
// (a counter word saturates near 88 atomics/us, so per-wave atomics from ten
// thousand waves of one frame serialise the whole stage).  Must be reached by
// every thread of the block.
__device__ __forceinline__ void casc_emit(const CascArgs& a, int frame, bool alive, int lvl, int n, int nWinR, float h)
{
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const unsigned long long mask = __ballot(alive);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
    {
        s_cnt[wv] = __popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = tot ? atomicAdd((a.last ? a.counts : a.qoutCount) + frame, tot) : 0;
    }
    __syncthreads();
    if (alive)
    {
        int base = s_base;
        for (int q = 0; q < wv; q++)
        {
            base += s_cnt[q];
        }
        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (a.last)
        {
            if (idx < a.maxHits)
            {
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = n / nWinR;
                hit.r = n - hit.c * nWinR;
                hit.score = h;
                a.hits[int64_t(frame) * a.maxHits + idx] = hit;
            }
        }
        else if (idx < a.qcap)
        {
            a.qout[int64_t(frame) * a.qcap + idx] = make_uint2((uint32_t(lvl) << 24) | uint32_t(n), __float_as_uint(h));
        }
    }
    __syncthreads(); // s_cnt / s_base are reused by the next grid-stride iteration
}

// Stage 0: windows enumerated in (level, c, r) order, level uniform per block.
template <int MODE>
__global__ void __launch_bounds__(256) k_cascade_first(CascArgs a)
{
    // 1-D grid, frame index fastest: consecutive hardware blocks belong to
    // different frames, so (i) concurrent workgroups spread their queue atomics
    // over nFrames counter words and (ii) with block b on XCD b % 8 each XCD works
    // on the same window blocks of 1/8 of the frames, whose overlapping 20x20xnC
    // footprints then share that XCD's L2.
    const int frame = blockIdx.x % a.nFrames;
    const int bx = blockIdx.x / a.nFrames;
    const int lvl = a.blockLevel[bx];
    const CascLevel L = a.levels[lvl];
    const int n = (bx - L.firstBlock) * blockDim.x + threadIdx.x;
    bool alive = n < L.nWin;
    const int c = alive ? n / L.nWinR : 0;
    const int r = alive ? n - c * L.nWinR : 0;
    const float* chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + (r * a.stride / a.shrink) + int64_t(c * a.stride / a.shrink) * L.hP;
    float h = 0.f;
    casc_eval<MODE>(a, chn, L.hP, L.hP * L.wP, L.nodeOff, h, alive);
    casc_emit(a, frame, alive, lvl, n, L.nWinR, h);
}

// Later stages: dense waves over the previous stage's survivor queue.
template <int MODE>
__global__ void __launch_bounds__(256) k_cascade_queue(CascArgs a)
{
    const int frame = blockIdx.x % a.nFrames;
    const int bq = blockIdx.x / a.nFrames, nbq = gridDim.x / a.nFrames;
    const int cnt = min(a.qinCount[frame], a.qcap);
    for (int base = bq * blockDim.x; base < cnt; base += nbq * blockDim.x)
    {
        const int i = base + threadIdx.x;
        bool alive = i < cnt;
        const uint2 e = alive ? a.qin[int64_t(frame) * a.qcap + i] : make_uint2(0u, 0u);
        const int lvl = int(e.x >> 24);
        const int n = int(e.x & 0xffffffu);
        const CascLevel L = a.levels[lvl];
        const int c = n / L.nWinR;
        const int r = n - c * L.nWinR;
        const float* chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + (r * a.stride / a.shrink) + int64_t(c * a.stride / a.shrink) * L.hP;
        float h = __uint_as_float(e.y);
        casc_eval<MODE>(a, chn, L.hP, L.hP * L.wP, L.nodeOff, h, alive);
        casc_emit(a, frame, alive, lvl, n, L.nWinR, h);
    }
}

// Tail stage [t0, nTrees): the few windows that are still alive (0.14 % on the
// headline workload) each need thousands of feature reads scattered over their
// own modelDsPad footprint.  With one lane per window every read is a separate
// cache line and nothing is reused, which makes the stage HBM-bound on 4-byte
// gathers.  Here one WAVE owns one window instead: the window's footprint
// (nChns*mW*mH floats = 16 KB for an 80x80 model — exactly the cids[] index
// space, acfDetect1.cpp:390-406, so a feature id addresses it directly) is
// copied to LDS once, then the 64 lanes evaluate 64 consecutive trees at a time
// from LDS and the leaf values are added to the running score strictly in tree
// order (a wave-uniform loop over lanes), stopping at the first h <= cascThr
// exactly like ParallelDetectionBody::evaluate (:123-138).
template <int MODE>
__global__ void __launch_bounds__(64) k_cascade_tail(CascArgs a)
{
    extern __shared__ float win[]; // nChns * mW * mH
    const int frame = blockIdx.x % a.nFrames;
    const int bq = blockIdx.x / a.nFrames, nbq = gridDim.x / a.nFrames;
    const int cnt = min(a.qinCount[frame], a.qcap);
    const int lane = threadIdx.x;
    const int cellsW = a.mW * a.mH, nFeat = a.nChns * cellsW;
    const float thrC = a.cascThr;
    for (int i = bq + a.qskip; i < cnt; i += nbq)
    {
        const uint2 e = a.qin[int64_t(frame) * a.qcap + i];
        const int lvl = int(e.x >> 24);
        const int n = int(e.x & 0xffffffu);
        const CascLevel L = a.levels[lvl];
        const int c = n / L.nWinR;
        const int r = n - c * L.nWinR;
        const float* chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + (r * a.stride / a.shrink) + int64_t(c * a.stride / a.shrink) * L.hP;
        const int area = L.hP * L.wP;
        __syncthreads();
        for (int f = lane; f < nFeat; f += 64)
        {
            const int z = f / cellsW, rem = f - z * cellsW;
            const int cc = rem / a.mH, rr = rem - cc * a.mH;
            win[f] = chn[z * area + cc * L.hP + rr];
        }
        __syncthreads();
        float h = __uint_as_float(e.y);
        bool alive = true;
        for (int tb = a.t0; tb < a.t1 && alive; tb += 64)
        {
            const int t = tb + lane;
            float hv = 0.f;
            if (t < a.t1)
            {
                const uint32_t offset = uint32_t(t) * uint32_t(a.nTreeNodes);
                uint32_t k = offset;
                if (MODE != 0)
                {
                    uint32_t k0 = 0;
                    const int D = (MODE == 2) ? 2 : a.treeDepth;
                    for (int q = 0; q < D; q++)
                    {
                        const float ftr = win[a.fids[k]];
                        k = (ftr < a.thrs[k]) ? 1 : 2;
                        k0 = k += k0 * 2;
                        k += offset;
                    }
                }
                else
                {
                    uint32_t k0 = offset;
                    while (a.child[k])
                    {
                        const float ftr = win[a.fids[k]];
                        k = (ftr < a.thrs[k]) ? 1 : 0;
                        k0 = k = a.child[k0] - k + offset;
                    }
                }
                hv = a.hs[k];
            }
            const int nt = min(64, a.t1 - tb);
            for (int q = 0; q < nt; q++)
            {
                h += __shfl(hv, q); // wave-uniform, tree order
                if (!(h > thrC))
                {
                    alive = false;
                    break;
                }
            }
        }
        if (alive && lane == 0)
        {
            const int idx = atomicAdd(a.counts + frame, 1);
            if (idx < a.maxHits)
            {
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = c;
                hit.r = r;
                hit.score = h;
                a.hits[int64_t(frame) * a.maxHits + idx] = hit;
            }
        }
    }
}

// The last stage of a fixed-depth model without the serial part of k_cascade_tail.  Which leaf a tree selects does not
// depend on the running score — only the early exit does (acfDetect1.cpp:123-138) — so the stage is (i) one byte per
// (window, tree), 4 * (leaf index), and (ii) an ordered scan with lanes = windows (k_tail_scan's, for 2^D leaves per tree).
// (i): a wave per queue entry, its footprint in LDS (feature ids address it directly), lanes = trees walking their D levels
// (getChild, :100-107).  k_cascade_tail adds the 64 leaves of a batch one lane at a time (1920 dependent steps per window
// and wave); here the additions of 64 WINDOWS run side by side in one wave of (ii).
__global__ void __launch_bounds__(64) k_tail_codesD(CascArgs a)
{
    extern __shared__ float win[]; // nChns * mW * mH
    const int frame = blockIdx.x % a.nFrames;
    const int bq = blockIdx.x / a.nFrames, nbq = gridDim.x / a.nFrames;
    const int cnt = min(min(a.qinCount[frame], a.qcap), a.codeCap);
    const int lane = threadIdx.x;
    const int cellsW = a.mW * a.mH, nFeat = a.nChns * cellsW;
    const int D = a.treeDepth, NN = (1 << D) - 1;
    for (int i = bq; i < cnt; i += nbq)
    {
        const uint2 e = a.qin[int64_t(frame) * a.qcap + i];
        const int lvl = int(e.x >> 24);
        const int n = int(e.x & 0xffffffu);
        const CascLevel L = a.levels[lvl];
        const int c = n / L.nWinR;
        const int r = n - c * L.nWinR;
        const float* chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + (r * a.stride / a.shrink) + int64_t(c * a.stride / a.shrink) * L.hP;
        const int area = L.hP * L.wP;
        __syncthreads();
        for (int f = lane; f < nFeat; f += 64)
        {
            const int z = f / cellsW, rem = f - z * cellsW;
            const int cc = rem / a.mH, rr = rem - cc * a.mH;
            win[f] = chn[z * area + cc * L.hP + rr];
        }
        __syncthreads();
        uint8_t* cp = a.codes + (int64_t(frame) * a.codeCap + i) * a.codePitch;
        // four batches of 64 trees side by side: a walk is D dependent steps of (node record from L2, feature from LDS), and
        // only independent walks hide each other's latency
        for (int tb = a.t0; tb < a.t1; tb += 256)
        {
            uint32_t off4[4], k4[4], k04[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                off4[u] = uint32_t(min(tb + 64 * u + lane, a.t1 - 1)) * uint32_t(a.nTreeNodes);
                k4[u] = off4[u];
                k04[u] = 0;
            }
            for (int q = 0; q < D; q++)
            {
                uint32_t fid[4];
                float thr[4], ftr[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    fid[u] = a.fids[k4[u]];
                    thr[u] = a.thrs[k4[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    ftr[u] = win[fid[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const uint32_t k = ((ftr[u] < thr[u]) ? 1u : 2u) + k04[u] * 2u;
                    k04[u] = k;
                    k4[u] = k + off4[u];
                }
            }
            // (lanes past the last tree write code 0: k_tail_scanD adds whole groups of 16 trees, and the rows of the trees
            // that do not exist hold -0.0f — `x + -0.0f` is x for every x)
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int j = tb - a.t0 + 64 * u + lane;
                if (j < a.codePitch)
                {
                    cp[j] = tb + 64 * u + lane < a.t1 ? uint8_t(4u * (k04[u] - uint32_t(NN))) : uint8_t(0);
                }
            }
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256) k_tail_scanD(CascArgs a)
{
    extern __shared__ float lds[]; // [nT][2^D] leaf values of trees [t0, t1)
    constexpr int NL = 1 << D, NN = NL - 1, LB = 4 * NL;
    const int frame = blockIdx.x % a.nFrames, chunk = blockIdx.x / a.nFrames;
    const int cntC = min(min(a.qinCount[frame], a.qcap), a.codeCap);
    if (chunk * 256 >= cntC)
    {
        return;
    }
    const int nT = a.t1 - a.t0, nT16 = (nT + 15) & ~15;
    for (int x = threadIdx.x; x < nT16 * NL; x += 256)
    {
        const int t = x / NL, j = x - t * NL;
        lds[x] = t < nT ? a.hs[int64_t(a.t0 + t) * a.nTreeNodes + NN + j] : -0.0f; // (padding: the identity of float addition)
    }
    __syncthreads();
    const int i = chunk * 256 + int(threadIdx.x);
    bool alive = i < cntC;
    const int ic = min(i, cntC - 1);
    const uint2 e = a.qin[int64_t(frame) * a.qcap + ic];
    const uint8_t* __restrict__ cp = a.codes + (int64_t(frame) * a.codeCap + ic) * a.codePitch;
    const float thrC = a.cascThr;
    float h = __uint_as_float(e.y);
    float m = h; // running minimum of the prefix scores
    const char* leafB = reinterpret_cast<const char*>(lds);
    // 16 trees (one 16-byte code load) per step, requested two steps ahead (a lane's codes are its own cache lines)
    uint4 w0 = *reinterpret_cast<const uint4*>(cp), w1 = *reinterpret_cast<const uint4*>(cp + min(16, a.codePitch - 16)), w2;
    bool done = false;
    int tb = 0;
#define TSD_STEP(W, T0)                                                                       \
    {                                                                                         \
        const char* lb = leafB + (T0) * LB;                                                   \
        _Pragma("unroll") for (int q = 0; q < 16; q++)                                        \
        {                                                                                     \
            const uint32_t cw = (q >> 2) == 0 ? W.x : ((q >> 2) == 1 ? W.y : ((q >> 2) == 2 ? W.z : W.w)); \
            const uint32_t off = (cw >> (8 * (q & 3))) & 0xffu;                               \
            h = h + *reinterpret_cast<const float*>(lb + q * LB + off);                       \
            asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));                                   \
        }                                                                                     \
    }
    while (tb < nT && !done)
    {
        w2 = *reinterpret_cast<const uint4*>(cp + min(tb + 32, a.codePitch - 16));
        TSD_STEP(w0, tb);
        tb += 16;
        w0 = w1;
        w1 = w2;
        if ((tb & 63) == 0)
        {
            alive = alive && (m > thrC) && (h > thrC);
            done = __ballot(alive) == 0ull; // every lane of the wave is rejected: nothing left to add
        }
    }
#undef TSD_STEP
    alive = alive && (m > thrC) && (h > thrC);
    const unsigned long long mask = __ballot(alive);
    if (mask)
    {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0)
        {
            base = atomicAdd(a.counts + frame, __popcll(mask));
        }
        base = __shfl(base, 0);
        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (alive && idx < a.maxHits)
        {
            const int lvl = int(e.x >> 24);
            const int n = int(e.x & 0xffffffu);
            const int nWinR = a.levels[lvl].nWinR;
            acf_hip_hit hit;
            hit.scale = lvl;
            hit.c = n / nWinR;
            hit.r = n - hit.c * nWinR;
            hit.score = h;
            a.hits[int64_t(frame) * a.maxHits + idx] = hit;
        }
    }
}

// ------------------------------------------------------------------------
// LDS-tiled cascade (depth-2 models, stride a multiple of shrink).
//
// A workgroup owns a tile of TR x TC windows of one level of one frame.  The
// tile's channel footprint — nChns planes of ((TC-1)*step + mW) columns by
// ((TR-1)*step + mH) rows — is read from the pyramid ONCE, with row-contiguous
// (coalesced) loads, into LDS; every feature fetch of every tree then comes
// from LDS.  With 32 x 16 windows of an 80x80 / 10-channel model that is 71 KB,
// two workgroups per CU, and each pyramid cell is fetched ~2.8x per frame in
// total instead of once per (window, tree node) touching it.
//
//   stage A  trees [b0,b1): one lane per window (lanes run along r, so a wave's LDS addresses are consecutive:
//            conflict-free), node records through the scalar unit;
//   sparse stages [b1,b2) [b2,b3) [b3,b4): items = survivors x trees, the score accumulated in tree order by a DPP chain;
//   stage E  the leaf codes of every remaining tree for the windows that reach the tail (k_tail_scan adds them up).
//   (k_cascade_tile3 below.)
//
// Scores are those of ParallelDetectionBody::evaluate (acfDetect1.cpp:123-138):
// every window adds the same leaves in the same order and stops at the first
// h <= cascThr.
// ------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* cptr4_t; // constant: loads through it may use the scalar unit

struct CascTile
{
    int16_t level, pad_;
    int16_t r0, c0; // first window row / column of the tile
};

static_assert(sizeof(CascTile) == 8, "k_cascade_tile3 reads a CascTile as two dwords");
static_assert(offsetof(CascLevel, nWinR) == 8 && offsetof(CascLevel, off) == 24 && offsetof(CascLevel, offR) == 40 && offsetof(CascLevel, pitchR) == 48,
    "k_cascade_tile3 reads a CascLevel as dwords");

// A tile's / a level's record through the scalar unit (read-only tables at workgroup-uniform addresses): the pooled tile kernels take
// their pointers inside an argument struct, where the compiler cannot prove the tables read-only and would issue vector loads +
// v_readfirstlane — two dependent L2 round trips at the head of every tile.  (As dwords: a CascTile's own alignment is 2.)
typedef const __attribute__((address_space(4))) uint32_t* cu32_k;
__device__ __forceinline__ CascTile load_tile_k(const CascTile* p)
{
    cu32_k tp = (cu32_k)(uintptr_t)p;
    const uint32_t w0 = tp[0], w1 = tp[1];
    CascTile T;
    T.level = int16_t(w0 & 0xffffu);
    T.pad_ = 0;
    T.r0 = int16_t(w1 & 0xffffu);
    T.c0 = int16_t(w1 >> 16);
    return T;
}
__device__ __forceinline__ CascLevel load_level_k(const CascLevel* p) // (the fields the tile kernels use)
{
    cu32_k lp = (cu32_k)(uintptr_t)p;
    CascLevel L;
    L.hP = int32_t(lp[0]);
    L.wP = int32_t(lp[1]);
    L.nWinR = int32_t(lp[2]);
    L.nWinC = int32_t(lp[3]);
    L.off = int64_t(uint64_t(lp[6]) | (uint64_t(lp[7]) << 32));
    L.offR = int64_t(uint64_t(lp[10]) | (uint64_t(lp[11]) << 32));
    L.pitchR = int32_t(lp[12]);
    return L;
}

struct __attribute__((aligned(16))) TreeNode
{
    uint32_t off[4]; // float offset of nodes 0,1,2 relative to the window's first cell; [3] unused
    float thr[4];
    float hs[4];     // leaves 3..6
};

struct TileGeom
{
    int32_t TR, TC, NW, W;     // window rows / columns per tile (TR * TC == 64 * NW * W), waves per workgroup, windows per lane
    int32_t step;              // stride / shrink, cells between adjacent windows
    int32_t rowsT, colsT;      // footprint rows / columns
    int32_t rowsP;             // LDS column stride: rowsT rounded up to 4 floats (16-byte fill chunks)
    int32_t tileFloats;        // nChns * colsT * rowsP
    uint32_t cpsMagic, colsMagic; // ceil(2^32 / (rowsP/4)), ceil(2^32 / colsT): exact q/d by mulhi for every chunk index (checked at plan time)
    int32_t b[5];              // stage boundaries b0=0 <= b1 <= b2 <= b3 <= b4 = tEnd
    int32_t winFloats;         // nChns * mW * mH (tail kernel's per-wave window)
    int32_t pooled;            // k_cascade_tile3: dense [0,b1) on every window, dense [b1,b2) on the workgroup's pooled survivors, b3 == b2, sparse [b2,b4)
    int32_t passW;             // k_cascade_tile3: windows per pass of the sparse stage (64 or 32: what the LDS budget allows)
    int32_t pitchC;            // k_cascade_tile3: bytes per window of the sparse stage's leaf codes (a multiple of 4 with pitchC / 4 odd, >= the trees padded to 16)
};

struct TileArgs
{
    const float* pyr;
    int64_t pyr_fs;
    const uint16_t* pyrR; // threshold-rank cells (host_plan.h): what the tile kernels reads instead of `pyr`
    int64_t pyrR_fs;
    const CascLevel* levels;
    const CascTile* tiles;
    int32_t nTiles, nFrames, nChns, mH, mW, nTrees;
    TileGeom g;
    const TreeNode* tileNodes;   // tile-layout offsets of every tree
    const uint32_t* tileNodesS;  // stage A of the tile kernels: 10 * aTB dwords per batch of aTB trees {off[aTB][3], thr[aTB][3], hs[aTB][4]}
    int32_t aTB;                 // trees per stage-A batch (4 or 8)
    const TreeNode* tailNodes;
    float cascThr;
    // tail queue [frame][qcap] {(level << 24) | window, h bits}; qcount[frame], qhead[frame]
    uint2* q;
    int32_t* qcount;
    int32_t* qhead;
    int32_t* tileNext; // k_cascade_tile3: [8] next tile of each XCD's range (zeroed before the launch)
    int32_t qcap;
    acf_hip_hit* hits;
    int32_t* counts;
    int32_t maxHits;
    int32_t debug; // timing experiments only (ACF_HIP_CASC_DEBUG): 1 = skip the tile fill, 2 = stop after the fill, 4 = phase stamps
    long long* stamps; // [block][8] s_memtime at phase boundaries (debug & 4)
    // k_cascade_tail3: per-wave leaf matrix [TAIL_G windows][tailPad trees] in global memory, LDS floats per wave
    float* tailScratch;
    int32_t tailPad, tailSlab;
    int32_t tailNodesLds; // the tail's node table fits in LDS next to the footprint slabs (floats reserved at the start of LDS, else 0)
    // stage E of the tile kernels / k_tail_scan: leaf codes [frame][codeCap][codePitch] bytes (4 * leaf index of every tail tree of a queued window)
    uint8_t* tailCodes;
    int32_t codeCap, codePitch;
};

// a tree's node record as one lane holds it (k_cascade_tail3: lanes = trees)
struct LaneNode
{
    uint4 o, tq, hq;
};

// Phase stamps and the timing exits of the tile kernels exist only in a build with -DACF_HIP_STAMPS (profiles/build_variant.sh):
// the shipped kernels carry no debug branches.
#ifdef ACF_HIP_STAMPS
#define TILE_STAMP(k)                                                          \
    if ((a.debug & 4) && threadIdx.x == 0)                                      \
    {                                                                          \
        a.stamps[int64_t(blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    }
#else
#define TILE_STAMP(k)
#endif

// ------------------------------------------------------------------------
// The cascade on LDS tiles.  Rounds 1-3 kept every wave on its own 64 windows from the dense trees to the sparse pieces
// (k_cascade_tile, k_cascade_tile2: deleted in round 5; DESIGN.md 3.1b has what was measured on them); what they established and
// k_cascade_tile3 keeps: stage A's node records through the scalar unit (a batch of four trees is 160 contiguous bytes read with
// s_load while the batch's feature reads are in flight), sparse stages as ITEMS = survivors x trees with the score accumulated in
// tree order, and the tail's leaf codes computed while the tile is still in LDS (stage E + k_tail_scan).
// ------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) uint32_t* cu32p_t;

// What a cell of the tile is.  CellF32: the fused pyramid's floats, node thresholds as float bits.  CellRank: 16-bit
// threshold ranks (host_plan.h, "threshold-rank cells"), node thresholds as rank indices: `rank(v) < k + 1` is `v < t_k`
// for every cell and every node of the model, so both forms take the same branch at every node, add the same leaves in
// the same order and stop at the same tree — in half the LDS and half the fill bytes.
struct CellF32
{
    typedef float cell_t;
    typedef float val_t;
    static constexpr int CPB = 4; // cells per 16-byte fill chunk
    static constexpr bool RANK = false;
    static __device__ __forceinline__ val_t thr(uint32_t bits) { return __uint_as_float(bits); }
};
struct CellRank
{
    typedef uint16_t cell_t;
    typedef uint32_t val_t;
    static constexpr int CPB = 8;
    static constexpr bool RANK = true;
    static __device__ __forceinline__ val_t thr(uint32_t bits) { return bits; }
};

// one tree at a time through the TreeNode table (stage A trees beyond the last full batch of four)
template <class CT>
__device__ __forceinline__ void tile_eval_s1(const typename CT::cell_t* win, const TreeNode* __restrict__ nodes, int t0, int t1, float thrC, float& h, bool& alive)
{
    typedef typename CT::val_t val_t;
    for (int t = t0; t < t1; t++)
    {
        cptr4_t np = (cptr4_t)(uintptr_t)(nodes + t);
        const u32x4 o = np[0], tq = np[1], hq = np[2];
        val_t f0 = val_t(win[o.x]), f1 = val_t(win[o.y]), f2 = val_t(win[o.z]);
        ACF_PIN_V(f0);
        ACF_PIN_V(f1);
        ACF_PIN_V(f2);
        const bool lt0 = f0 < CT::thr(tq.x);
        const val_t fc = lt0 ? f1 : f2;
        const val_t th1 = CT::thr(lt0 ? tq.y : tq.z);
        const bool lt1 = fc < th1;
        const float hv = __uint_as_float(lt0 ? (lt1 ? hq.x : hq.y) : (lt1 ? hq.z : hq.w));
        const float hn = h + hv;
        h = hn;
        alive = alive && (hn > thrC);
    }
}

// Survivors of the last tile stage -> hits (model exhausted) or the frame's tail queue; returns the queue slot / hit index
// of this lane's entry (-1: not emitted).  One global atomic per wave.  (The destination fields are passed one by one:
// k_cascade_tile3 reads them from the kernarg segment at the call.)
struct EmitDst
{
    acf_hip_hit* hits;
    int32_t* counts;
    uint2* q;
    int32_t* qcount;
    int32_t maxHits, qcap;
};
__device__ __forceinline__ int tile_emit3(const EmitDst& d, bool final_, int frame, bool alive, int lvl, int n, int nWinR, float h)
{
    const unsigned long long mask = __ballot(alive);
    if (!mask)
    {
        return -1;
    }
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0)
    {
        base = atomicAdd((final_ ? d.counts : d.qcount) + frame, __popcll(mask));
    }
    base = __shfl(base, 0);
    int idx = -1;
    if (alive)
    {
        idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (final_)
        {
            if (idx < d.maxHits)
            {
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = n / nWinR;
                hit.r = n - hit.c * nWinR;
                hit.score = h;
                d.hits[int64_t(frame) * d.maxHits + idx] = hit;
            }
        }
        else if (idx < d.qcap)
        {
            d.q[int64_t(frame) * d.qcap + idx] = make_uint2((uint32_t(lvl) << 24) | uint32_t(n), __float_as_uint(h));
        }
    }
    return idx;
}

template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) // lane l <- lane l - N of its 16-lane row (0 where there is none)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_row_bcast15(float v) // every lane of row r <- lane 15 of row r-1 (row 0 keeps its own)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xf, 0xf, false));
}

// a <- a + (leaf of lane j of this row), j = 0..15 in order, in lane 15 of every row; m <- min of the prefixes.  The DPP
// operand is the leaf, not the running sum, so the chain is 16 dependent v_add and nothing else.
#define ACF_ROW_STEP(N)                                                     \
    a = a + dpp_row_shr<N>(leaf);                                           \
    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(a));
__device__ __forceinline__ void row_chain(float leaf, float& a, float& m)
{
    ACF_ROW_STEP(15) ACF_ROW_STEP(14) ACF_ROW_STEP(13) ACF_ROW_STEP(12) ACF_ROW_STEP(11) ACF_ROW_STEP(10) ACF_ROW_STEP(9) ACF_ROW_STEP(8)
    ACF_ROW_STEP(7) ACF_ROW_STEP(6) ACF_ROW_STEP(5) ACF_ROW_STEP(4) ACF_ROW_STEP(3) ACF_ROW_STEP(2) ACF_ROW_STEP(1)
    a = a + leaf;
    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(a));
}
#undef ACF_ROW_STEP

// ------------------------------------------------------------------------
// k_cascade_tile3: the tile kernel with the survivors POOLED over the workgroup.
//
// Round 3 kept every wave on its own 64 windows: 32 dense trees for each of them (the mean window needs 14), then
// sparse pieces on the wave's ~2 survivors whose rounds cost a wave ~300 instructions however few of its lanes hold an item —
// together 849 VALU instructions per wave, of which the kernel's time is the issue time (profiles/r03_pmc_sq_*).  Here:
//  A1  trees [0, b1) (16): lanes = the wave's own windows (tile_eval_p: records through the scalar unit, leaves
//      added under EXEC).  Survivors {window, score} go to ONE list of the workgroup (a ballot and one LDS atomic per wave).
//  A2  trees [b1, b2) (16..32): the same dense evaluation with lanes = list entries: ceil(n1 / 64) waves run it, the others
//      go to the barrier and leave the SIMD's issue slots to the CU's other workgroups.
//  S   trees [b2, b4) (32..128), items = survivors x trees, every thread one tree (node in registers), 8 / 4 windows per
//      round: two dependent LDS reads give the leaf's code byte (4 * leaf index, as stage E's), written to codes[window][tree];
//      then ONE wave, lanes = windows, adds the leaves in tree order — code byte -> leaf table in LDS -> h += leaf, min over the
//      prefixes — which is evaluate()'s chain (acfDetect1.cpp:123-138) for up to 64 windows at once.
//  E   leaf codes (round 2.s stage E: of the tail trees for the windows that enter the tail queue).
// A window's score is the same chain of f32 additions in the same order in every stage (A: v_add under EXEC per tree; S: the
// one-wave chain; rows of -0.0f pad the leaf table to 16 trees, the identity of float addition).
// LDS: [leaf table 2 KB][tile cells][R1: list 1 = h[NWIN] f32 + tag[NWIN] u16, later the codes 64 x pitchC][R2: list 2, same
// form; its head becomes stage E's {tag, slot} list in place].
// ------------------------------------------------------------------------
#define TILE3_LEAF_BYTES 2048 // 128 trees x 4 leaves x 4 bytes
typedef const __attribute__((address_space(4))) TileArgs* tile_args_k; // the kernel's argument block in the kernarg segment

// Stage A of k_cascade_tile3: a batch of four trees per scalar load, a tree's three compares inside the asm block of its four leaf adds, so that
// a tree's wave masks live for seven instructions instead of a batch (24 SGPRs fewer across the loop: the kernel's later
// phases keep their scalars in registers instead of v_writelane / v_readlane round trips, which are VALU instructions).
template <class CT>
__device__ __forceinline__ void tile_eval_p(const typename CT::cell_t* win, const uint32_t* __restrict__ tab, int nBatches, float thrC, float& h, bool& alive)
{
    typedef typename CT::val_t val_t;
    constexpr int TB = 4;
    cu32p_t p = (cu32p_t)(uintptr_t)tab;
    uint32_t o[3 * TB];
#pragma unroll
    for (int i = 0; i < 3 * TB; i++)
    {
        o[i] = p[i];
    }
    const unsigned long long execAll = __builtin_amdgcn_read_exec(); // every lane of the wave is here (callers: wave-uniform control flow only)
    float hMin = __builtin_inff();
    for (int b = 0; b < nBatches; b++)
    {
        val_t f[3 * TB];
#pragma unroll
        for (int i = 0; i < 3 * TB; i++)
        {
            f[i] = val_t(win[o[i]]);
        }
        cu32p_t pb = p + 10 * TB * b;
        uint32_t th[3 * TB], hv4[4 * TB];
#pragma unroll
        for (int i = 0; i < 3 * TB; i++)
        {
            th[i] = pb[3 * TB + i];
        }
#pragma unroll
        for (int i = 0; i < 4 * TB; i++)
        {
            hv4[i] = pb[6 * TB + i];
        }
#pragma unroll
        for (int i = 0; i < 3 * TB; i++)
        {
            ACF_PIN_V(f[i]);
        }
        cu32p_t pn = p + 10 * TB * min(b + 1, nBatches - 1);
#pragma unroll
        for (int i = 0; i < 3 * TB; i++)
        {
            o[i] = pn[i];
        }
#pragma unroll
        for (int g = 0; g < TB; g += 2)
        {
            float h1, h2;
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                const int t = g + q;
                float hOut;
                const float hIn = q == 0 ? h : h1;
                unsigned long long m0, mA, mB;
                if (CT::RANK)
                {
                    asm volatile("v_cmp_gt_u32 %[m0], %[t0], %[f0]\n\t"
                                 "v_cmp_gt_u32 %[mA], %[t1], %[f1]\n\t"
                                 "v_cmp_gt_u32 %[mB], %[t2], %[f2]\n\t"
                                 "s_and_b64 exec, %[m0], %[mA]\n\t"
                                 "v_add_f32 %[o], %[A], %[i]\n\t"
                                 "s_andn2_b64 exec, %[m0], %[mA]\n\t"
                                 "v_add_f32 %[o], %[B], %[i]\n\t"
                                 "s_andn2_b64 exec, %[mB], %[m0]\n\t"
                                 "v_add_f32 %[o], %[C], %[i]\n\t"
                                 "s_nor_b64 exec, %[m0], %[mB]\n\t"
                                 "v_add_f32 %[o], %[D], %[i]\n\t"
                                 "s_mov_b64 exec, %[ex]"
                                 : [o] "=&v"(hOut), [m0] "=&s"(m0), [mA] "=&s"(mA), [mB] "=&s"(mB)
                                 : [i] "v"(hIn), [f0] "v"(f[3 * t]), [f1] "v"(f[3 * t + 1]), [f2] "v"(f[3 * t + 2]), [t0] "s"(th[3 * t]), [t1] "s"(th[3 * t + 1]),
                                 [t2] "s"(th[3 * t + 2]), [A] "s"(hv4[4 * t]), [B] "s"(hv4[4 * t + 1]), [C] "s"(hv4[4 * t + 2]), [D] "s"(hv4[4 * t + 3]), [ex] "s"(execAll)
                                 : "scc");
                }
                else
                {
                    asm volatile("v_cmp_gt_f32 %[m0], %[t0], %[f0]\n\t"
                                 "v_cmp_gt_f32 %[mA], %[t1], %[f1]\n\t"
                                 "v_cmp_gt_f32 %[mB], %[t2], %[f2]\n\t"
                                 "s_and_b64 exec, %[m0], %[mA]\n\t"
                                 "v_add_f32 %[o], %[A], %[i]\n\t"
                                 "s_andn2_b64 exec, %[m0], %[mA]\n\t"
                                 "v_add_f32 %[o], %[B], %[i]\n\t"
                                 "s_andn2_b64 exec, %[mB], %[m0]\n\t"
                                 "v_add_f32 %[o], %[C], %[i]\n\t"
                                 "s_nor_b64 exec, %[m0], %[mB]\n\t"
                                 "v_add_f32 %[o], %[D], %[i]\n\t"
                                 "s_mov_b64 exec, %[ex]"
                                 : [o] "=&v"(hOut), [m0] "=&s"(m0), [mA] "=&s"(mA), [mB] "=&s"(mB)
                                 : [i] "v"(hIn), [f0] "v"(f[3 * t]), [f1] "v"(f[3 * t + 1]), [f2] "v"(f[3 * t + 2]), [t0] "s"(th[3 * t]), [t1] "s"(th[3 * t + 1]),
                                 [t2] "s"(th[3 * t + 2]), [A] "s"(hv4[4 * t]), [B] "s"(hv4[4 * t + 1]), [C] "s"(hv4[4 * t + 2]), [D] "s"(hv4[4 * t + 3]), [ex] "s"(execAll)
                                 : "scc");
                }
                if (q == 0)
                {
                    h1 = hOut;
                }
                else
                {
                    h2 = hOut;
                }
            }
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(hMin) : "v"(h1), "v"(h2));
            h = h2; // a rejected window's score is never read again
        }
    }
    alive = alive && (hMin > thrC);
}

template <int NW, class CT>
__global__ void __launch_bounds__(NW * 64) k_cascade_tile3(TileArgs a)
{
    typedef typename CT::cell_t cell_t;
    typedef typename CT::val_t val_t;
    constexpr int CPB = CT::CPB;
    constexpr int NT = NW * 64;
    extern __shared__ float lds[];
    __shared__ int s_n[4]; // entries in list 1, list 2, (unused), stage E's list
    float* leafT = lds;
    cell_t* tileF = reinterpret_cast<cell_t*>(reinterpret_cast<char*>(lds) + TILE3_LEAF_BYTES);
    const int NWIN = a.g.TR * a.g.TC;
    // list entries: {score bits, tag | window offset << 16}: tag = column * TR + row of the window in the tile, offset = its first cell in the tile
    char* r1 = reinterpret_cast<char*>(tileF) + size_t(a.g.tileFloats) * sizeof(cell_t);
    const int passW = a.g.passW;
    const int r1Bytes = (max(NWIN * 8, passW * a.g.pitchC) + 15) & ~15;
    uint2* l1 = reinterpret_cast<uint2*>(r1);
    uint8_t* codes = reinterpret_cast<uint8_t*>(r1);
    uint2* l2 = reinterpret_cast<uint2*>(r1 + r1Bytes);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // Persistent workgroups: the grid is what the CUs hold at once; a workgroup draws tiles from the counter of its XCD (one
    // contiguous range of frame-major tiles per XCD, as k_cascade_tile), the next one while the current tile is being filled.
    const int64_t total = int64_t(a.nTiles) * a.nFrames;
    const int perX = int((total + 7) >> 3);
    const int xcd = blockIdx.x & 7;
    __shared__ int s_next[2]; // (two slots: a wave that is late reading tile n's successor never meets tile n + 1's write)
    const bool persist = a.tileNext != nullptr; // else: one tile per workgroup, blockIdx.x -> tile
    int li = int(blockIdx.x >> 3);
    if (persist) // (a kernel argument: workgroup-uniform)
    {
        if (tid == 0)
        {
            s_next[0] = atomicAdd(a.tileNext + xcd, 1);
        }
        __syncthreads();
        li = __builtin_amdgcn_readfirstlane(s_next[0]);
    }
    int par = 1;
    const int step = a.g.step, rowsP = a.g.rowsP, TR = a.g.TR;
    const int b1 = a.g.b[1], b2 = a.g.b[2], tEnd = a.g.b[4];
    // the sparse stage: this thread's tree (TLp = 32 / 64 / 128 threads per window), its node in registers; the stage's leaf table
    const int Ts = tEnd - b2, TsPad = (Ts + 15) & ~15;
    const int tlShift = TsPad <= 32 ? 5 : (TsPad <= 64 ? 6 : 7);
    const int pos = tid & ((1 << tlShift) - 1);
    uint32_t so0 = 0, so1 = 0, so2 = 0, st0 = 0, st1 = 0, st2 = 0;
    if (Ts > 0)
    {
        const uint4* np = reinterpret_cast<const uint4*>(a.tileNodes + b2 + min(pos, Ts - 1));
        const uint4 o = np[0], tq = np[1];
        so0 = o.x, so1 = o.y, so2 = o.z;
        st0 = tq.x, st1 = tq.y, st2 = tq.z;
    }
    bool leavesDone = Ts <= 0; // (the leaf table is copied once, behind the first tile's fill requests: its loads ride on the fill's latency)
    for (;;)
    {
    const int64_t id = int64_t(xcd) * perX + li;
    if (li >= perX || id >= total) // (workgroup-uniform)
    {
        break;
    }
    int liNext = perX;
    if (tid == 0 && persist)
    {
        liNext = atomicAdd(a.tileNext + xcd, 1); // (returns during the fill)
    }
    // tile and level records through the scalar unit (read-only tables, workgroup-uniform addresses): two short dependent s_loads
    // where vector loads + v_readfirstlane were two L2 round trips; everything the window test needs arrives with them, so
    // nothing is re-read behind the fill's barrier
    const int frame = int(id / a.nTiles);
    const CascTile T = load_tile_k(a.tiles + (id - int64_t(frame) * a.nTiles));
    const int lvl = T.level;
    const CascLevel L = load_level_k(a.levels + lvl);
    const int nWinR = L.nWinR;
    if (tid < 4)
    {
        s_n[tid] = 0;
    }
    TILE_STAMP(0);
    // ---- fill (k_cascade_tile's: 16-byte LDS-DMA chunks, everything in flight at once)
    {
        const int colsT = a.g.colsT;
        const int gr0 = T.r0 * step, gc0 = T.c0 * step;
        const int colPitch = CT::RANK ? L.pitchR : L.hP;
        const int area = colPitch * L.wP;
        const cell_t* __restrict__ src0 = (CT::RANK ? reinterpret_cast<const cell_t*>(a.pyrR) + int64_t(frame) * a.pyrR_fs + L.offR
                                                    : reinterpret_cast<const cell_t*>(a.pyr) + int64_t(frame) * a.pyr_fs + L.off) + gr0;
        // (the BUFFER form of the LDS-DMA: behind a global_load_lds the compiler waits for every request in flight before the
        // kernel's next LDS access — here the write of the next tile's index — so that the leaf table's loads below were requested
        // only after the fill had ARRIVED: two memory round trips in sequence at the head of every one-tile workgroup)
        const srd_t fsrd = make_srd(src0, int64_t(a.nChns) * area * int64_t(sizeof(cell_t)));
        const int colsValid = min(colsT, L.wP - gc0);
        const uint32_t cps = uint32_t(rowsP) / uint32_t(CPB);
        const uint32_t nChunks = uint32_t(a.nChns * colsT) * cps;
        const int ccMax = colsValid - 1;
        for (uint32_t q0 = uint32_t(wv) * 64u; q0 < nChunks; q0 += NW * 64u)
        {
            const uint32_t q = q0 + lane;
            if (q < nChunks)
            {
                const uint32_t seg = __umulhi(q, a.g.cpsMagic);
                const uint32_t j = q - seg * cps;
                const uint32_t z = __umulhi(seg, a.g.colsMagic);
                const int cc = int(seg - z * uint32_t(colsT));
                const uint32_t soff = z * uint32_t(area) + uint32_t(gc0 + min(cc, ccMax)) * uint32_t(colPitch) + uint32_t(CPB) * j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(fsrd, (lptr_t)(tileF + uint32_t(CPB) * q0), 16, soff * uint32_t(sizeof(cell_t)), 0, 0, 0);
            }
        }
    }
    // stage A1's window of this lane: wave w takes the window columns w, w + NW, ... (conflict-free feature reads)
    const int r_l = lane % TR, c_l = (lane / TR) * NW + wv;
    const bool aliveA1 = (T.r0 + r_l) < nWinR && (T.c0 + c_l) < L.nWinC && lane < (64 / TR) * TR;
    if (!leavesDone)
    {
        for (int t = tid; t < TsPad; t += NT)
        {
            float4 hv = make_float4(-0.f, -0.f, -0.f, -0.f); // rows past the last tree: h + -0.0f == h for every h
            if (t < Ts)
            {
                hv = *reinterpret_cast<const float4*>(a.tileNodes[b2 + t].hs);
            }
            // (four float stores, not one float4 store: type-based alias analysis is what tells the compiler that an LDS access
            // does not touch what the fill's requests write — a float4 access is ordered behind them, i.e. drains the fill first)
            leafT[4 * t] = hv.x;
            leafT[4 * t + 1] = hv.y;
            leafT[4 * t + 2] = hv.z;
            leafT[4 * t + 3] = hv.w;
        }
        leavesDone = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0)
    {
        s_next[par] = liNext; // (read after this tile's last barrier; written here, behind the wait: an LDS write in the fill's shadow would drain it)
    }
    __syncthreads();
    TILE_STAMP(1);

    const float thrC = a.cascThr;
    // survivors of the model's last tile tree: hits (model exhausted) or the frame's tail queue + stage E's list {slot, tag | offset}
    auto finish = [&](tile_args_k A, bool alive, uint32_t tw, float h) {
        const int tag = int(tw & 0xffffu);
        const int rl = tag % TR, cl = tag / TR;
        const bool lastAll = tEnd == A->nTrees;
        const EmitDst dst{ A->hits, A->counts, A->q, A->qcount, A->maxHits, A->qcap };
        const int slot = tile_emit3(dst, lastAll, frame, alive, lvl, (T.c0 + cl) * nWinR + (T.r0 + rl), nWinR, h);
        if (!lastAll && A->codeCap > 0)
        {
            const unsigned long long m = __ballot(alive);
            if (m)
            {
                int base = 0;
                if (lane == 0)
                {
                    base = atomicAdd(&s_n[3], __popcll(m));
                }
                base = __shfl(base, 0);
                if (alive)
                {
                    l2[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(uint32_t(slot), tw);
                }
            }
        }
    };
    auto append = [&](bool alive, uint32_t tw, float h, uint2* list, int* cnt) {
        const unsigned long long m = __ballot(alive);
        if (m)
        {
            int base = 0;
            if (lane == 0)
            {
                base = atomicAdd(cnt, __popcll(m));
            }
            base = __shfl(base, 0);
            if (alive)
            {
                list[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(__float_as_uint(h), tw);
            }
        }
    };
    // dense trees [t0, t1) of one window per lane (t0 a multiple of four: the plan gives this kernel batches of four trees)
    auto dense = [&](const cell_t* win, int t0, int t1, float& h, bool& alive) {
        const int nb = (t1 - t0) / 4;
        if (nb > 0)
        {
            tile_eval_p<CT>(win, a.tileNodesS + size_t(t0 / 4) * 40, nb, thrC, h, alive);
        }
        tile_eval_s1<CT>(win, a.tileNodes, t0 + nb * 4, t1, thrC, h, alive);
    };

    // ---- A1: lanes = windows, trees [0, b1).  Wave w takes the window columns w, w + NW, ... (conflict-free feature reads)
    {
        bool alive = aliveA1;
        float h = 0.f;
        const uint32_t woff = uint32_t((min(c_l, a.g.TC - 1) * step) * rowsP + r_l * step);
        dense(tileF + woff, 0, b1, h, alive);
        const uint32_t tw = uint32_t(c_l * TR + r_l) | (woff << 16);
        if (b1 == tEnd)
        {
            finish((tile_args_k)__builtin_amdgcn_kernarg_segment_ptr(), alive, tw, h);
        }
        else
        {
            append(alive, tw, h, b1 == b2 ? l2 : l1, b1 == b2 ? &s_n[1] : &s_n[0]);
        }
    }
    TILE_STAMP(2);
    __syncthreads();
    TILE_STAMP(3);
    // From here on the arguments are re-read from the kernarg segment where they are used (scalar loads): values kept alive across
    // stage A1's loop would be spilled to VGPR lanes and fetched back with one VALU instruction each.
    tile_args_k ak = (tile_args_k)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ak));
    // ---- A2: lanes = entries of list 1, trees [b1, b2)
    if (b1 < b2)
    {
        const int n1 = s_n[0];
        for (int e0 = wv * 64; e0 < n1; e0 += NT) // (n1 <= NT: one chunk per wave at most)
        {
            const int e = e0 + lane;
            bool alive = e < n1;
            const uint2 en = l1[alive ? e : e0];
            float h = __uint_as_float(en.x);
            dense(tileF + (en.y >> 16), b1, b2, h, alive);
            if (b2 == tEnd)
            {
                finish(ak, alive, en.y, h);
            }
            else
            {
                append(alive, en.y, h, l2, &s_n[1]);
            }
        }
        __syncthreads();
    }
    TILE_STAMP(4);
    const bool lastAll = tEnd == ak->nTrees;
    const bool wantE = !lastAll && ak->codeCap > 0;
    // ---- S: trees [b2, tEnd) for list 2, passW windows per pass
    if (b2 < tEnd)
    {
        const int n2 = s_n[1];
        const int wpr = NT >> tlShift; // windows per round
        const int pitchC = ak->g.pitchC;
        for (int p0 = 0; p0 < n2; p0 += passW)
        {
            const int nP = min(passW, n2 - p0);
            // items: two rounds side by side (their LDS round trips overlap)
            for (int wi = tid >> tlShift; wi < nP; wi += 2 * wpr)
            {
                const int wj = wi + wpr;
                const bool two = wj < nP;
                const cell_t* winA = tileF + (l2[p0 + wi].y >> 16);
                const cell_t* winB = tileF + (l2[p0 + (two ? wj : wi)].y >> 16);
                const val_t fA = val_t(winA[so0]), fB = val_t(winB[so0]);
                const bool ltA = fA < CT::thr(st0), ltB = fB < CT::thr(st0);
                const val_t cA = val_t(winA[ltA ? so1 : so2]), cB = val_t(winB[ltB ? so1 : so2]);
                const bool l1A = cA < CT::thr(ltA ? st1 : st2), l1B = cB < CT::thr(ltB ? st1 : st2);
                if (pos < TsPad)
                {
                    codes[wi * pitchC + pos] = pos < Ts ? uint8_t((ltA ? 0 : 8) + (l1A ? 0 : 4)) : uint8_t(0);
                    if (two)
                    {
                        codes[wj * pitchC + pos] = pos < Ts ? uint8_t((ltB ? 0 : 8) + (l1B ? 0 : 4)) : uint8_t(0);
                    }
                }
            }
            __syncthreads();
            // the ordered chain, FOUR lanes per window (a wave takes 16 windows; waves 0 .. 3 a pass of 64): lane j of a window
            // fetches the leaves of trees 4j .. 4j + 3 of every group of 16 — one dword of code bytes, four table reads — and the
            // window's score is added up in tree order with the leaves broadcast inside the quad (DPP quad_perm), identically
            // in its four lanes: evaluate()'s additions in evaluate()'s order.  (One lane per window read 16 + 16 times per
            // group and its wave ran alone: 2.7k of a tile's 18k cycles.)  Next group's leaves and the code dword after that
            // are requested before a group is added.
            constexpr int CH = NW >= 4 ? 1 : 4 / NW; // chunks of 16 windows per wave (a pass is at most 64 windows)
            bool aliveS[CH];
            int slotS[CH];
            uint32_t tagS[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ch++)
            {
                aliveS[ch] = false;
                slotS[ch] = -1;
                tagS[ch] = 0;
                const int w0 = (ch * NW + wv) * 16;
                if (w0 >= nP || w0 >= 64) // (wave-uniform)
                {
                    continue;
                }
                const int wl = w0 + (lane >> 2), j = lane & 3;
                const bool valid = wl < nP;
                const uint2 en = l2[p0 + (valid ? wl : 0)];
                float h = __uint_as_float(en.x);
                float hMin = __builtin_inff();
                const char* crow = reinterpret_cast<const char*>(codes) + (valid ? wl : 0) * pitchC + 4 * j;
                const char* lt = reinterpret_cast<const char*>(leafT) + 64 * j; // tree 16 g + 4 j + k: row at 256 g + 64 j + 16 k
                const int nG = TsPad >> 4;
                uint32_t cw = *reinterpret_cast<const uint32_t*>(crow);
                float lf[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    lf[k] = *reinterpret_cast<const float*>(lt + 16 * k + ((cw >> (8 * k)) & 0xffu));
                }
                cw = *reinterpret_cast<const uint32_t*>(crow + 16 * min(1, nG - 1));
                for (int g = 0; g < nG; g++)
                {
                    float cur[4], nx[4];
                    const char* ltN = lt + 256 * min(g + 1, nG - 1);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        cur[k] = lf[k];
                        nx[k] = *reinterpret_cast<const float*>(ltN + 16 * k + ((cw >> (8 * k)) & 0xffu));
                    }
                    const uint32_t cwN = *reinterpret_cast<const uint32_t*>(crow + 16 * min(g + 2, nG - 1));
                    __builtin_amdgcn_sched_barrier(0);
#define ACF_QUAD_BCAST(x, q) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), (q) * 0x55, 0xf, 0xf, true))
#define ACF_CHAIN_QUAD(q)                                                            \
    {                                                                                \
        const float h1 = h + ACF_QUAD_BCAST(cur[0], q);                              \
        const float h2 = h1 + ACF_QUAD_BCAST(cur[1], q);                             \
        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(hMin) : "v"(h1), "v"(h2));            \
        const float h3 = h2 + ACF_QUAD_BCAST(cur[2], q);                             \
        const float h4 = h3 + ACF_QUAD_BCAST(cur[3], q);                             \
        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(hMin) : "v"(h3), "v"(h4));            \
        h = h4;                                                                      \
    }
                        ACF_CHAIN_QUAD(0)
                        ACF_CHAIN_QUAD(1)
                        ACF_CHAIN_QUAD(2)
                        ACF_CHAIN_QUAD(3)
#undef ACF_CHAIN_QUAD
#undef ACF_QUAD_BCAST
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        lf[k] = nx[k];
                    }
                    cw = cwN;
                }
                aliveS[ch] = valid && j == 0 && hMin > thrC;
                tagS[ch] = en.y;
                if (__ballot(aliveS[ch])) // (most chunks end with no window alive: none of the index arithmetic then)
                {
                    const int tag = int(en.y & 0xffffu);
                    const int rl = tag % TR, cl = tag / TR;
                    const EmitDst dst{ ak->hits, ak->counts, ak->q, ak->qcount, ak->maxHits, ak->qcap };
                    slotS[ch] = tile_emit3(dst, lastAll, frame, aliveS[ch], lvl, (T.c0 + cl) * nWinR + (T.r0 + rl), nWinR, h);
                }
            }
            __syncthreads(); // (the next pass rewrites the codes; every chain wave has read its entries of list 2)
            if (wantE)
            {
                // stage E's list, in place at the head of list 2: (entries so far) + (survivors of this pass) <= p0 + nP, this
                // pass's entries are in registers, and the next pass reads from p0 + passW on
#pragma unroll
                for (int ch = 0; ch < CH; ch++)
                {
                    const unsigned long long m = __ballot(aliveS[ch]);
                    if (m)
                    {
                        int base = 0;
                        if (lane == 0)
                        {
                            base = atomicAdd(&s_n[3], __popcll(m));
                        }
                        base = __shfl(base, 0);
                        if (aliveS[ch])
                        {
                            l2[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(uint32_t(slotS[ch]), tagS[ch]);
                        }
                    }
                }
            }
        }
    }
    TILE_STAMP(5);
    __syncthreads(); // (stage E's list is complete; every wave is done with the codes and the lists' other uses)
    li = __builtin_amdgcn_readfirstlane(s_next[par]);
    par ^= 1;
    if (!wantE)
    {
        continue;
    }
    // ---- E: leaf codes of every tail tree for the windows now in the tail queue (stage E over one list)
    const int nTail = s_n[3];
#ifdef ACF_HIP_STAMPS
    if ((a.debug & 4) && threadIdx.x == 0)
    {
        a.stamps[int64_t(blockIdx.x) * 8 + 7] = (long long)s_n[0] | ((long long)s_n[1] << 16) | ((long long)nTail << 32);
    }
#endif
    if (nTail != 0)
    {
        const int nTrees = ak->nTrees, codeCap = ak->codeCap, codePitch = ak->codePitch;
        const TreeNode* __restrict__ nodes = ak->tileNodes + tEnd;
        uint8_t* __restrict__ codesG = ak->tailCodes + int64_t(frame) * codeCap * codePitch + lane;
        const int nT = nTrees - tEnd, nB = (nT + 63) >> 6;
        for (int b0 = wv; b0 < nB; b0 += 4 * NW)
        {
            uint32_t o0[4], o1[4], o2[4];
            val_t t0[4], t1[4], t2[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int b = min(b0 + k * NW, nB - 1);
                const uint4* np = reinterpret_cast<const uint4*>(nodes + min(b * 64 + lane, nT - 1));
                const uint4 o = np[0], tq = np[1];
                o0[k] = o.x;
                o1[k] = o.y;
                o2[k] = o.z;
                t0[k] = CT::thr(tq.x);
                t1[k] = CT::thr(tq.y);
                t2[k] = CT::thr(tq.z);
            }
            for (int s = 0; s < nTail; s++)
            {
                const uint2 en = l2[s];
                const int slot = int(en.x);
                if (slot < 0 || slot >= codeCap)
                {
                    continue; // no code row: k_cascade_tail3 / k_cascade_tail_rank takes this entry
                }
                const cell_t* win = tileF + (en.y >> 16);
                uint8_t* __restrict__ row = codesG + int64_t(slot) * codePitch;
                val_t f0[4], fc[4];
                bool lt0[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    f0[k] = val_t(win[o0[k]]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    lt0[k] = f0[k] < t0[k];
                    fc[k] = val_t(win[lt0[k] ? o1[k] : o2[k]]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const bool lt1 = fc[k] < (lt0[k] ? t1[k] : t2[k]);
                    if (b0 + k * NW < nB) // wave-uniform
                    {
                        row[(b0 + k * NW) * 64] = uint8_t((lt0[k] ? 0 : 8) + (lt1 ? 0 : 4));
                    }
                }
            }
        }
    }
    __syncthreads(); // (the next tile's fill rewrites the cells stage E reads)
    }
}

// Copy one window's footprint (nChns*mW*mH floats, the cids[] index space) from the pyramid level into a wave's LDS
// slab.  run = z * mW + cc  ->  win[run * mH + rr].
// ------------------------------------------------------------------------
// k_cascade_tileD: the first trees of a fixed-depth model OTHER than depth 2 (acfDetect1.cpp:201-228 dispatches depth
// 1..8 through one body) on LDS tiles.  Same tiles, fill and window mapping as the depth-2 tile kernel (float cells); stage A is
// generalised to depth D: a tree's 2^D - 1 node features are all read (the walk would be D dependent LDS round trips), the
// compares give wave masks, the 2^D leaf masks are ANDs along the paths (scalar unit), and every leaf is added under EXEC
// to the lanes of its mask — the additions and their order are evaluate()'s (:123-138).  Nodes are in heap order (node k's
// children 2k + 1 for `ftr < thr`, 2k + 2 otherwise, getChild :100-107); leaf j counts the paths left to right.
// Survivors of trees [0, t1) go to the staged path's queue ({(level << 24) | window, h}: k_cascade_queue / k_cascade_tail
// finish them from the float pyramid) or, when t1 is the model's last tree, to the hits.
// Records: per batch of TB trees {off[TB][NN], thr[TB][NN], hs[TB][NL]} dwords, NN = 2^D - 1, NL = 2^D (host: buildTileSet).
// ------------------------------------------------------------------------
struct TileDArgs
{
    const float* pyr;
    int64_t pyr_fs;
    const CascLevel* levels;
    const CascTile* tiles;
    int32_t nTiles, nFrames, nChns, nBatches;
    TileGeom g;
    const uint32_t* nodesD;
    float cascThr;
    int32_t last;      // the stage ends the model: survivors are hits
    uint2* qout;       // [frame][qcap]
    int32_t* qoutCount;
    int32_t qcap;
    acf_hip_hit* hits;
    int32_t* counts;
    int32_t maxHits;
    // k_cascade_tile3D (pooled survivors, see k_cascade_tile3): the model's heap-ordered nodes [tree][nTreeNodes] — tile offsets of
    // the internal nodes, thresholds, leaves (hs: the last 2^D of a tree's entries) —, the leaf codes of the tail trees
    const uint32_t* tileOff;
    const float* thrs;       // float cells: the thresholds; rank cells (CellRank): their rank indices as uint32 bit patterns
    const float* hs;
    const uint16_t* pyrR;    // CellRank: the rank pyramid (CascLevel::offR / pitchR)
    int64_t pyrR_fs;
    int32_t nTrees, nTreeNodes;
    uint8_t* codes; // [frame][codeCap][codePitch]: 4 * leaf index of every tree >= g.b[4] for the first codeCap queue entries
    int32_t codeCap, codePitch;
    int32_t* tileNext; // [8] per-XCD tile counters (persistent workgroups), or nullptr: one tile per workgroup
};

template <int D, int TB, class CT = CellF32>
__device__ __forceinline__ void tile_eval_d(const typename CT::cell_t* win, const uint32_t* __restrict__ tab, int nBatches, float thrC, float& h, bool& alive)
{
    constexpr int NN = (1 << D) - 1, NL = 1 << D, REC = TB * (2 * NN + NL);
    cu32p_t p = (cu32p_t)(uintptr_t)tab;
    uint32_t o[TB * NN];
#pragma unroll
    for (int i = 0; i < TB * NN; i++)
    {
        o[i] = p[i];
    }
    const unsigned long long execAll = __builtin_amdgcn_read_exec(); // (callers: wave-uniform control flow only)
    float hMin = __builtin_inff();
    for (int b = 0; b < nBatches; b++)
    {
        typename CT::val_t f[TB * NN];
#pragma unroll
        for (int i = 0; i < TB * NN; i++)
        {
            f[i] = typename CT::val_t(win[o[i]]);
        }
        cu32p_t pb = p + REC * b;
        uint32_t th[TB * NN], hv[TB * NL];
#pragma unroll
        for (int i = 0; i < TB * NN; i++)
        {
            th[i] = pb[TB * NN + i];
        }
#pragma unroll
        for (int i = 0; i < TB * NL; i++)
        {
            hv[i] = pb[2 * TB * NN + i];
        }
#pragma unroll
        for (int i = 0; i < TB * NN; i++)
        {
            ACF_PIN_V(f[i]);
        }
        cu32p_t pn = p + REC * min(b + 1, nBatches - 1);
#pragma unroll
        for (int i = 0; i < TB * NN; i++)
        {
            o[i] = pn[i];
        }
#pragma unroll
        for (int t = 0; t < TB; t++)
        {
            unsigned long long m[NN];
#pragma unroll
            for (int k = 0; k < NN; k++)
            {
                m[k] = __builtin_amdgcn_ballot_w64(f[t * NN + k] < CT::thr(th[t * NN + k]));
            }
            float hOut = h;
#pragma unroll
            for (int j = 0; j < NL; j++)
            {
                // the path of leaf j: bit (D - 1 - l) of j is the branch taken at level l (0: ftr < thr)
                unsigned long long mj = execAll;
                int k = 0;
#pragma unroll
                for (int l = 0; l < D; l++)
                {
                    const int bit = (j >> (D - 1 - l)) & 1;
                    mj &= bit ? ~m[k] : m[k];
                    k = 2 * k + 1 + bit;
                }
                asm volatile("s_mov_b64 exec, %[m]\n\t"
                             "v_add_f32 %[o], %[L], %[i]\n\t"
                             "s_mov_b64 exec, %[ex]"
                             : [o] "+v"(hOut)
                             : [i] "v"(h), [m] "s"(mj), [L] "s"(hv[t * NL + j]), [ex] "s"(execAll));
            }
            h = hOut;
            hMin = fminf(hMin, h); // (a window is rejected as soon as one prefix is <= cascThr)
        }
    }
    alive = alive && (hMin > thrC);
}

template <int NW, int D, int TB>
__global__ void __launch_bounds__(NW * 64) k_cascade_tileD(TileDArgs a)
{
    extern __shared__ float lds[];
    float* tileF = lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t total = int64_t(a.nTiles) * a.nFrames;
    const int64_t perX = (total + 7) >> 3;
    const int64_t id = int64_t(blockIdx.x & 7) * perX + (blockIdx.x >> 3); // one contiguous range of frame-major tiles per XCD
    if (id >= total || (blockIdx.x >> 3) >= perX)
    {
        return;
    }
    const int frame = int(id / a.nTiles);
    const CascTile T = a.tiles[id - int64_t(frame) * a.nTiles];
    const int lvl = T.level;
    const CascLevel L = a.levels[lvl];
    const int step = a.g.step, rowsP = a.g.rowsP, colsT = a.g.colsT;
    const int gr0 = T.r0 * step, gc0 = T.c0 * step;
    const int colPitch = L.hP;
    const int area = colPitch * L.wP;
    const float* __restrict__ src0 = a.pyr + int64_t(frame) * a.pyr_fs + L.off + gr0;
    const int colsValid = min(colsT, L.wP - gc0);
    // ---- fill (16-byte LDS-DMA chunks, everything in flight at once)
    {
        const uint32_t cps = uint32_t(rowsP) / 4u;
        const uint32_t nChunks = uint32_t(a.nChns * colsT) * cps;
        const int ccMax = colsValid - 1;
        for (uint32_t q0 = uint32_t(wv) * 64u; q0 < nChunks; q0 += NW * 64u)
        {
            const uint32_t q = q0 + lane;
            if (q < nChunks)
            {
                const uint32_t seg = __umulhi(q, a.g.cpsMagic);
                const uint32_t j = q - seg * cps;
                const uint32_t z = __umulhi(seg, a.g.colsMagic);
                const int cc = int(seg - z * uint32_t(colsT));
                const uint32_t soff = z * uint32_t(area) + uint32_t(gc0 + min(cc, ccMax)) * uint32_t(colPitch) + 4u * j;
                __builtin_amdgcn_global_load_lds((gptr_t)(src0 + soff), (lptr_t)(tileF + 4u * q0), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int r_l = lane % a.g.TR, c_l = (lane / a.g.TR) * NW + wv;
    const int wr = T.r0 + r_l;
    bool alive = wr < L.nWinR && (T.c0 + c_l) < L.nWinC && lane < (64 / a.g.TR) * a.g.TR;
    float h = 0.f;
    const float* win = tileF + (min(c_l, a.g.TC - 1) * step) * rowsP + r_l * step;
    tile_eval_d<D, TB>(win, a.nodesD, a.nBatches, a.cascThr, h, alive);
    // survivors: one global atomic per wave
    const unsigned long long mask = __ballot(alive);
    if (!mask)
    {
        return;
    }
    int base = 0;
    if (lane == 0)
    {
        base = atomicAdd((a.last ? a.counts : a.qoutCount) + frame, __popcll(mask));
    }
    base = __builtin_amdgcn_readfirstlane(base);
    if (alive)
    {
        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        const int n = (T.c0 + c_l) * L.nWinR + wr;
        if (a.last)
        {
            if (idx < a.maxHits)
            {
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = T.c0 + c_l;
                hit.r = wr;
                hit.score = h;
                a.hits[int64_t(frame) * a.maxHits + idx] = hit;
            }
        }
        else if (idx < a.qcap)
        {
            a.qout[int64_t(frame) * a.qcap + idx] = make_uint2((uint32_t(lvl) << 24) | uint32_t(n), __float_as_uint(h));
        }
    }
}

// ------------------------------------------------------------------------
// k_cascade_tile3D: k_cascade_tile3's pooled stages for the fixed depths other than 2 (acfDetect1.cpp:201-228 runs depth 1..8
// through one body), on float cells.  A1 trees [0, b1) on every window and A2 trees [b1, b2) on the workgroup's pooled
// survivors with tile_eval_d (all 2^D - 1 node compares of a tree as wave masks, the 2^D leaf adds under EXEC); S trees
// [b2, b4) as leaf codes (a thread per (window, tree): the D-level walk with the tree's nodes in registers, picked by
// select trees — no node record is fetched inside the walk) and ONE wave's ordered chain through the leaf table in LDS;
// E the codes of the tail trees [b4, nTrees) for the windows that enter the tail queue (k_tail_scanD adds them up).
// Before this kernel the depths 1, 3, 4 took trees [32, 128) from global memory (k_cascade_queue) and re-read every tail
// window's footprint for its codes (k_tail_codesD): 86 us per 1080p frame at depth 3, 278 us at depth 4 against depth 2's 25.
// ------------------------------------------------------------------------
template <int N, class T>
__device__ __forceinline__ T sel_pow2(const T* a, uint32_t j)
{
    if constexpr (N == 1)
    {
        return a[0];
    }
    else
    {
        const T lo = sel_pow2<N / 2, T>(a, j), hi = sel_pow2<N / 2, T>(a + N / 2, j);
        return (j & uint32_t(N / 2)) ? hi : lo;
    }
}

// the leaf a window reaches in one tree: o[] / th[] = the tree's internal nodes in heap order (node k's children 2k + 1 for
// ftr < thr, 2k + 2 otherwise: getChild, acfDetect1.cpp:100-107); returns the leaf index 0 .. 2^D - 1, left to right
template <int D, int L, class CT>
__device__ __forceinline__ uint32_t walk_from(const typename CT::cell_t* win, const uint32_t (&o)[(1 << D) - 1], const uint32_t (&th)[(1 << D) - 1], uint32_t p)
{
    if constexpr (L == D)
    {
        return p;
    }
    else
    {
        // level L: p holds the L decisions so far, the node is the p-th of the level's 2^L (heap index 2^L - 1 + p)
        const uint32_t off = sel_pow2<(1 << L), uint32_t>(o + ((1 << L) - 1), p);
        const uint32_t thr = sel_pow2<(1 << L), uint32_t>(th + ((1 << L) - 1), p); // (threshold bits: CT::thr)
        return walk_from<D, L + 1, CT>(win, o, th, 2u * p + (typename CT::val_t(win[off]) < CT::thr(thr) ? 0u : 1u));
    }
}
template <int D, class CT>
__device__ __forceinline__ uint32_t walk_tree(const typename CT::cell_t* win, const uint32_t (&o)[(1 << D) - 1], const uint32_t (&th)[(1 << D) - 1])
{
    return walk_from<D, 0, CT>(win, o, th, 0u);
}

template <int NW, int D, int TB, class CT>
__global__ void __launch_bounds__(NW * 64) k_cascade_tile3D(TileDArgs a)
{
    typedef typename CT::cell_t cell_t;
    constexpr int CPB = CT::CPB;
    constexpr int NT = NW * 64, NN = (1 << D) - 1, NL = 1 << D, LB = 4 * NL, REC = TB * (2 * NN + NL);
    constexpr int LEAF_BYTES = 128 * LB;
    extern __shared__ float lds[];
    __shared__ int s_n[4];
    __shared__ int s_next[2];
    float* leafT = lds;
    cell_t* tileF = reinterpret_cast<cell_t*>(reinterpret_cast<char*>(lds) + LEAF_BYTES);
    const int NWIN = a.g.TR * a.g.TC;
    char* r1 = reinterpret_cast<char*>(tileF) + size_t(a.g.tileFloats) * sizeof(cell_t);
    const int passW = a.g.passW;
    const int r1Bytes = (max(NWIN * 8, passW * a.g.pitchC) + 15) & ~15;
    uint2* l1 = reinterpret_cast<uint2*>(r1);
    uint8_t* codes = reinterpret_cast<uint8_t*>(r1);
    uint2* l2 = reinterpret_cast<uint2*>(r1 + r1Bytes);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t total = int64_t(a.nTiles) * a.nFrames;
    const int perX = int((total + 7) >> 3);
    const int xcd = blockIdx.x & 7;
    const bool persist = a.tileNext != nullptr;
    int li = int(blockIdx.x >> 3);
    if (persist) // (a kernel argument: workgroup-uniform)
    {
        if (tid == 0)
        {
            s_next[0] = atomicAdd(a.tileNext + xcd, 1);
        }
        __syncthreads();
        li = __builtin_amdgcn_readfirstlane(s_next[0]);
    }
    int par = 1;
    const int step = a.g.step, rowsP = a.g.rowsP, TR = a.g.TR;
    const int b1 = a.g.b[1], b2 = a.g.b[2], tEnd = a.g.b[4];
    const bool lastAll = tEnd == a.nTrees;
    const bool wantE = !lastAll && a.codeCap > 0;
    const float thrC = a.cascThr;
    // the sparse stage: this thread's tree, its nodes in registers; the stage's leaf table
    const int Ts = tEnd - b2, TsPad = (Ts + 15) & ~15;
    const int tlShift = TsPad <= 32 ? 5 : (TsPad <= 64 ? 6 : 7);
    const int pos = tid & ((1 << tlShift) - 1);
    uint32_t so[NN], sth[NN];
    {
        const int64_t q = int64_t(b2 + min(pos, max(Ts, 1) - 1)) * a.nTreeNodes;
#pragma unroll
        for (int k = 0; k < NN; k++)
        {
            so[k] = Ts > 0 ? a.tileOff[q + k] : 0u;
            sth[k] = Ts > 0 ? __float_as_uint(a.thrs[q + k]) : 0u;
        }
    }
    bool leavesDone = false; // (the leaf table is copied once, behind the first tile's fill requests)
    for (;;)
    {
        const int64_t id = int64_t(xcd) * perX + li;
        if (li >= perX || id >= total) // (workgroup-uniform)
        {
            break;
        }
        int liNext = perX;
        if (tid == 0 && persist)
        {
            liNext = atomicAdd(a.tileNext + xcd, 1);
        }
        const int frame = int(id / a.nTiles);
        const CascTile T = load_tile_k(a.tiles + (id - int64_t(frame) * a.nTiles));
        const int lvl = T.level;
        const CascLevel L = load_level_k(a.levels + lvl);
        if (tid < 4)
        {
            s_n[tid] = 0;
        }
        // ---- fill (k_cascade_tileD's)
        {
            const int colsT = a.g.colsT;
            const int gr0 = T.r0 * step, gc0 = T.c0 * step;
            const int colPitch = CT::RANK ? L.pitchR : L.hP;
            const int area = colPitch * L.wP;
            const cell_t* __restrict__ src0 = (CT::RANK ? reinterpret_cast<const cell_t*>(a.pyrR) + int64_t(frame) * a.pyrR_fs + L.offR
                                                        : reinterpret_cast<const cell_t*>(a.pyr) + int64_t(frame) * a.pyr_fs + L.off) + gr0;
            const int colsValid = min(colsT, L.wP - gc0);
            const uint32_t cps = uint32_t(rowsP) / uint32_t(CPB);
            const uint32_t nChunks = uint32_t(a.nChns * colsT) * cps;
            const int ccMax = colsValid - 1;
            for (uint32_t q0 = uint32_t(wv) * 64u; q0 < nChunks; q0 += NW * 64u)
            {
                const uint32_t q = q0 + lane;
                if (q < nChunks)
                {
                    const uint32_t seg = __umulhi(q, a.g.cpsMagic);
                    const uint32_t j = q - seg * cps;
                    const uint32_t z = __umulhi(seg, a.g.colsMagic);
                    const int cc = int(seg - z * uint32_t(colsT));
                    const uint32_t soff = z * uint32_t(area) + uint32_t(gc0 + min(cc, ccMax)) * uint32_t(colPitch) + uint32_t(CPB) * j;
                    __builtin_amdgcn_global_load_lds((gptr_t)(src0 + soff), (lptr_t)(tileF + uint32_t(CPB) * q0), 16, 0, 0);
                }
            }
        }
        if (tid == 0)
        {
            s_next[par] = liNext;
        }
        if (!leavesDone)
        {
            for (int x = tid; x < TsPad * NL; x += NT)
            {
                const int t = x / NL, j = x - t * NL;
                leafT[x] = t < Ts ? a.hs[int64_t(b2 + t) * a.nTreeNodes + NN + j] : -0.0f; // (padding: h + -0.0f == h for every h)
            }
            leavesDone = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nWinR = L.nWinR;
        auto emit = [&](bool alive, uint32_t tw, float h) -> int {
            const int tag = int(tw & 0xffffu);
            const int rl = tag % TR, cl = tag / TR;
            const EmitDst dst{ a.hits, a.counts, a.qout, a.qoutCount, a.maxHits, a.qcap };
            return tile_emit3(dst, lastAll, frame, alive, lvl, (T.c0 + cl) * nWinR + (T.r0 + rl), nWinR, h);
        };
        auto finish = [&](bool alive, uint32_t tw, float h) {
            const int slot = emit(alive, tw, h);
            if (wantE)
            {
                const unsigned long long m = __ballot(alive);
                if (m)
                {
                    int base = 0;
                    if (lane == 0)
                    {
                        base = atomicAdd(&s_n[3], __popcll(m));
                    }
                    base = __shfl(base, 0);
                    if (alive)
                    {
                        l2[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(uint32_t(slot), tw);
                    }
                }
            }
        };
        auto append = [&](bool alive, uint32_t tw, float h, uint2* list, int* cnt) {
            const unsigned long long m = __ballot(alive);
            if (m)
            {
                int base = 0;
                if (lane == 0)
                {
                    base = atomicAdd(cnt, __popcll(m));
                }
                base = __shfl(base, 0);
                if (alive)
                {
                    list[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(__float_as_uint(h), tw);
                }
            }
        };
        // ---- A1
        {
            const int r_l = lane % TR, c_l = (lane / TR) * NW + wv;
            bool alive = (T.r0 + r_l) < nWinR && (T.c0 + c_l) < L.nWinC && lane < (64 / TR) * TR;
            float h = 0.f;
            const uint32_t woff = uint32_t((min(c_l, a.g.TC - 1) * step) * rowsP + r_l * step);
            tile_eval_d<D, TB, CT>(tileF + woff, a.nodesD, b1 / TB, thrC, h, alive);
            const uint32_t tw = uint32_t(c_l * TR + r_l) | (woff << 16);
            if (b1 == tEnd)
            {
                finish(alive, tw, h);
            }
            else
            {
                append(alive, tw, h, b1 == b2 ? l2 : l1, b1 == b2 ? &s_n[1] : &s_n[0]);
            }
        }
        __syncthreads();
        // ---- A2
        if (b1 < b2)
        {
            const int n1 = s_n[0];
            for (int e0 = wv * 64; e0 < n1; e0 += NT)
            {
                const int e = e0 + lane;
                bool alive = e < n1;
                const uint2 en = l1[alive ? e : e0];
                float h = __uint_as_float(en.x);
                tile_eval_d<D, TB, CT>(tileF + (en.y >> 16), a.nodesD + size_t(b1 / TB) * REC, (b2 - b1) / TB, thrC, h, alive);
                if (b2 == tEnd)
                {
                    finish(alive, en.y, h);
                }
                else
                {
                    append(alive, en.y, h, l2, &s_n[1]);
                }
            }
            __syncthreads();
        }
        // ---- S
        if (b2 < tEnd)
        {
            const int n2 = s_n[1];
            const int wpr = NT >> tlShift;
            const int pitchC = a.g.pitchC;
            int nE = 0;
            for (int p0 = 0; p0 < n2; p0 += passW)
            {
                const int nP = min(passW, n2 - p0);
                for (int wi = tid >> tlShift; wi < nP; wi += wpr)
                {
                    const cell_t* win = tileF + (l2[p0 + wi].y >> 16);
                    const uint32_t leaf = walk_tree<D, CT>(win, so, sth);
                    if (pos < TsPad)
                    {
                        codes[wi * pitchC + pos] = pos < Ts ? uint8_t(4u * leaf) : uint8_t(0);
                    }
                }
                __syncthreads();
                if (wv == 0)
                {
                    const bool valid = lane < nP;
                    const uint2 en = l2[p0 + (valid ? lane : 0)];
                    float h = __uint_as_float(en.x);
                    float hMin = __builtin_inff();
                    const uint8_t* crow = codes + (valid ? lane : 0) * pitchC;
                    const char* lt = reinterpret_cast<const char*>(leafT);
                    uint32_t cb[16];
#pragma unroll
                    for (int k = 0; k < 16; k++)
                    {
                        cb[k] = crow[k];
                    }
                    for (int t = 0; t < TsPad; t += 16)
                    {
                        float lf[16];
#pragma unroll
                        for (int k = 0; k < 16; k++)
                        {
                            lf[k] = *reinterpret_cast<const float*>(lt + LB * (t + k) + cb[k]);
                        }
                        const int tn = min(t + 16, TsPad - 16);
#pragma unroll
                        for (int k = 0; k < 16; k++)
                        {
                            cb[k] = crow[tn + k];
                        }
#pragma unroll
                        for (int k = 0; k < 16; k += 2)
                        {
                            const float h1 = h + lf[k];
                            const float h2 = h1 + lf[k + 1];
                            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(hMin) : "v"(h1), "v"(h2));
                            h = h2;
                        }
                    }
                    const bool alive = valid && hMin > thrC;
                    const int slot = emit(alive, en.y, h);
                    if (wantE)
                    {
                        const unsigned long long m = __ballot(alive);
                        __builtin_amdgcn_wave_barrier();
                        if (alive)
                        {
                            l2[nE + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(uint32_t(slot), en.y);
                        }
                        nE += __popcll(m);
                    }
                }
                __syncthreads();
            }
            if (wantE && tid == 0)
            {
                s_n[3] = nE;
            }
        }
        __syncthreads();
        li = __builtin_amdgcn_readfirstlane(s_next[par]);
        par ^= 1;
        if (!wantE)
        {
            continue;
        }
        // ---- E: the codes of the tail trees for this tile's queue entries: lanes = trees, one 64-tree batch per wave at a time
        const int nTail = s_n[3];
        if (nTail != 0)
        {
            const int nT = a.nTrees - tEnd, nB = (nT + 63) >> 6;
            for (int b = wv; b < nB; b += NW)
            {
                uint32_t eo[NN], eth[NN];
                const int64_t q = int64_t(tEnd + min(b * 64 + lane, nT - 1)) * a.nTreeNodes;
#pragma unroll
                for (int k = 0; k < NN; k++)
                {
                    eo[k] = a.tileOff[q + k];
                    eth[k] = __float_as_uint(a.thrs[q + k]);
                }
                for (int s = 0; s < nTail; s++)
                {
                    const uint2 en = l2[s];
                    const int slot = int(en.x);
                    if (slot < 0 || slot >= a.codeCap)
                    {
                        continue; // no code row: k_cascade_tail takes this entry
                    }
                    const uint32_t leaf = walk_tree<D, CT>(tileF + (en.y >> 16), eo, eth);
                    a.codes[(int64_t(frame) * a.codeCap + slot) * a.codePitch + b * 64 + lane] = b * 64 + lane < nT ? uint8_t(4u * leaf) : uint8_t(0);
                }
            }
        }
        __syncthreads(); // (the next tile's fill rewrites the cells stage E reads)
    }
}

struct TailFill
{
    int SUB, sub, rr, rrc, nRuns;
    bool lact;
    uint32_t cpsMagic, mwMagic;
};

__device__ __forceinline__ TailFill tail_fill_setup(const TileArgs& a, int lane)
{
    TailFill t;
    const int mH = a.mH, mW = a.mW;
    // lane -> (sub-row of this pass, row offset) for the 4-byte copy: SUB runs of mH floats per pass
    t.SUB = max(1, min(64 / mH, mW)); // <= mW: one conditional wrap per step
    t.sub = lane / mH;
    t.rr = lane - t.sub * mH;
    t.lact = t.sub < t.SUB;
    t.rrc = t.lact ? t.rr : 0;
    t.nRuns = a.nChns * mW;
    t.cpsMagic = uint32_t(((uint64_t(1) << 32) + uint32_t(max(mH >> 2, 1)) - 1) / uint32_t(max(mH >> 2, 1)));
    t.mwMagic = uint32_t(((uint64_t(1) << 32) + uint32_t(mW) - 1) / uint32_t(mW));
    return t;
}

__device__ __forceinline__ void tail_fill(const TileArgs& a, float* win, const float* __restrict__ chn, int hP, int area, int lane, const TailFill& t)
{
    const int mH = a.mH, mW = a.mW;
    const int SUB = t.SUB, sub = t.sub, rrc = t.rrc, nRuns = t.nRuns;
    const bool lact = t.lact;
    const uint32_t cpsMagic = t.cpsMagic, mwMagic = t.mwMagic;
    struct
    {
        int hP;
    } L{ hP };
        // copy: run = z * mW + cc  ->  win[run * mH + rr], straight into LDS by LDS-DMA (no VGPR round trip); nothing
        // waits between instructions, so the whole footprint is in flight at once
        if ((mH & 3) == 0)
        {
            // 16-byte chunks: chunk q = floats [4q, 4q+4) of the window; 64 chunks (1 KB) per instruction
            const uint32_t cps = uint32_t(mH) >> 2, nChunks = uint32_t(nRuns) * cps;
            for (uint32_t q0 = 0; q0 < nChunks; q0 += 64u)
            {
                const uint32_t q = q0 + lane;
                if (q < nChunks)
                {
                    const uint32_t run = cps == 1 ? q : __umulhi(q, cpsMagic); // exact for q, cps < 2^16 (the magic of 1 is 2^32)
                    const uint32_t j = q - run * cps;
                    const uint32_t z = mW == 1 ? run : __umulhi(run, mwMagic);
                    const uint32_t cc = run - z * uint32_t(mW);
                    __builtin_amdgcn_global_load_lds((gptr_t)(chn + (z * uint32_t(area) + cc * uint32_t(L.hP) + 4u * j)), (lptr_t)(win + 4u * q0), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        else if (mH <= 64)
        {
            int z = 0, cc = lact ? sub : 0;
            const int zMax = a.nChns - 1;
            for (int base = 0; base < nRuns; base += SUB) // wave-uniform trip count
            {
                if (lact && base + sub < nRuns)
                {
                    __builtin_amdgcn_global_load_lds((gptr_t)(chn + (uint32_t(min(z, zMax)) * uint32_t(area) + uint32_t(cc * L.hP + rrc))),
                        (lptr_t)(win + base * mH), 4, 0, 0);
                }
                cc += SUB;
                if (cc >= mW)
                {
                    cc -= mW;
                    z++;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        else
        {
            for (int f = lane; f < a.g.winFloats; f += 64)
            {
                const int run = f / mH, r2 = f - run * mH;
                const int z = run / mW, cc = run - z * mW;
                win[f] = chn[int64_t(z) * area + cc * L.hP + r2];
            }
        }
}

// Tail stage for queue entries WITHOUT leaf codes (beyond codeCap per frame, or tiles whose geometry keeps stage E off):
// two phases per wave.
//
//   phase 1  lanes = trees.  The wave takes TAIL_G windows from the frame's queue; for each one it copies the
//            footprint to its LDS slab and walks ALL remaining trees 64 at a time, writing the leaf values
//            hs[k] to its private leaf matrix [window][tree] in global memory (256-byte rows, stays in L2 /
//            Infinity Cache: it is rewritten by the same wave every round).  No score is involved, so the
//            batches are independent and overlap.
//   phase 2  lanes = windows.  Lane w adds window w's leaf values to its score strictly in tree order,
//            h = h + hs (evaluate(), acfDetect1.cpp:123-138), 64 trees per step through a [TAIL_G][68]-float
//            LDS transposition tile (global rows in, one ds_read_b128 per 4 trees out); a lane dies when any
//            prefix is <= cascThr.  The add order per window is the reference's, so scores are bit-identical;
//            trees evaluated past a window's rejection point only cost phase-1 time.
constexpr int TAIL_G = 16;
constexpr int TAIL_PITCH = 68;

template <int NW, bool NODES_LDS>
__global__ void __launch_bounds__(NW * 64) k_cascade_tail3(TileArgs a)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* win = lds + a.tailNodesLds + wv * a.tailSlab;
    const int frame = blockIdx.x % a.nFrames;
    const int cnt = min(a.qcount[frame], a.qcap);
    if (a.qhead[frame] >= cnt) // nothing left in this frame's queue (k_tail_scan took it all): skip the node-table preload
    {
        return;
    }
    const int tEnd = a.g.b[4];
    const int nT = a.nTrees - tEnd;
    const int pad = a.tailPad;
    float* __restrict__ S = a.tailScratch + (int64_t(blockIdx.x) * NW + wv) * int64_t(TAIL_G) * pad;
    const TailFill tf = tail_fill_setup(a, lane);
    // Node table of the tail in LDS when it fits: every window walks the same trees, and an L2 round trip per
    // 64-tree batch (~1 us under load) was the whole cost of phase 1 when the nodes were read from global memory.
    // (a template parameter, not a run-time flag: with both node paths in the loop the compiler put a full
    // s_waitcnt vmcnt(0) at the top of every step, i.e. one store round trip per 64 trees)
    constexpr bool nodesLds = NODES_LDS;
    const TreeNode* __restrict__ nodeBase = a.tailNodes + tEnd;
    if (nodesLds)
    {
        const uint4* src = reinterpret_cast<const uint4*>(nodeBase);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < nT * 3; i += NW * 64)
        {
            dst[i] = src[i];
        }
        __syncthreads();
    }
    const float thrC = a.cascThr;
    for (;;)
    {
        int i0 = 0;
        if (lane == 0)
        {
            i0 = atomicAdd(a.qhead + frame, TAIL_G);
        }
        i0 = __builtin_amdgcn_readfirstlane(__shfl(i0, 0));
        if (i0 >= cnt)
        {
            break;
        }
        const int nW = min(TAIL_G, cnt - i0);
        // lane w < nW keeps window w's queue entry for phase 2
        const uint2 mine = a.q[int64_t(frame) * a.qcap + i0 + min(lane, nW - 1)];
        // ---- phase 1
        for (int k = 0; k < nW; k++)
        {
            const uint32_t ex = uint32_t(__builtin_amdgcn_readlane(int(mine.x), k));
            const int lvl = int(ex >> 24);
            const int n = int(ex & 0xffffffu);
            const CascLevel L = a.levels[lvl];
            const int c = n / L.nWinR;
            const int r = n - c * L.nWinR;
            const float* __restrict__ chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + r * a.g.step + int64_t(c * a.g.step) * L.hP;
            __builtin_amdgcn_wave_barrier(); // the previous window's feature reads are done (LDS ops of a wave are in order)
            tail_fill(a, win, chn, L.hP, L.hP * L.wP, lane, tf);
            __builtin_amdgcn_wave_barrier();
            float* __restrict__ row = S + k * pad;
            // four 64-tree batches per step: their node reads, root reads, child reads and stores are independent, so
            // the four LDS latency chains overlap (one batch per step was 760 cycles of exposed latency per batch)
            for (int tb = 0; tb < nT; tb += 256)
            {
                LaneNode nd[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const int t = min(tb + 64 * j + lane, nT - 1); // batches past the end: clamped duplicates, stored into the row's padding or skipped
                    if (nodesLds)
                    {
                        const uint4* np = reinterpret_cast<const uint4*>(lds) + 3 * t;
                        nd[j].o = np[0];
                        nd[j].tq = np[1];
                        nd[j].hq = np[2];
                    }
                    else
                    {
                        const uint4* np = reinterpret_cast<const uint4*>(nodeBase + t);
                        nd[j].o = np[0];
                        nd[j].tq = np[1];
                        nd[j].hq = np[2];
                    }
                }
                float f0[4], fc[4];
                bool lt0[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    f0[j] = win[nd[j].o.x];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    lt0[j] = f0[j] < __uint_as_float(nd[j].tq.x);
                    fc[j] = win[lt0[j] ? nd[j].o.y : nd[j].o.z];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const float th1 = __uint_as_float(lt0[j] ? nd[j].tq.y : nd[j].tq.z);
                    const bool lt1 = fc[j] < th1;
                    const float leaf = __uint_as_float(lt0[j] ? (lt1 ? nd[j].hq.x : nd[j].hq.y) : (lt1 ? nd[j].hq.z : nd[j].hq.w));
                    if (tb + 64 * j < pad)
                    {
                        row[tb + 64 * j + lane] = leaf;
                    }
                }
            }
        }
        // ---- phase 2.  Other lanes of this wave wrote the rows read below.  Workgroup scope is enough: the stores went
        // through this CU's write-through L1, which the loads below also use (an agent-scope fence would write back and
        // invalidate the XCD's whole L2 on gfx942/950 — measured 0.3 ms per 64 frames)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float* tile = win;
        float h = __uint_as_float(mine.y);
        bool alive = lane < nW;
        const int wl = lane & (TAIL_G - 1);
        float nx[TAIL_G];
#pragma unroll
        for (int k = 0; k < TAIL_G; k++)
        {
            nx[k] = S[k * pad + lane];
        }
        for (int tb = 0; tb < nT; tb += 64)
        {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < TAIL_G; k++)
            {
                tile[k * TAIL_PITCH + lane] = nx[k];
            }
            if (tb + 64 < nT)
            {
#pragma unroll
                for (int k = 0; k < TAIL_G; k++)
                {
                    nx[k] = S[k * pad + tb + 64 + lane];
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int nt = min(64, nT - tb);
            float m = h;
            if (nt == 64)
            {
#pragma unroll
                for (int q = 0; q < 16; q++)
                {
                    const float4 x = *reinterpret_cast<const float4*>(tile + wl * TAIL_PITCH + 4 * q);
                    h = h + x.x;
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
                    h = h + x.y;
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
                    h = h + x.z;
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
                    h = h + x.w;
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
                }
            }
            else
            {
                for (int q = 0; q < nt; q++)
                {
                    h = h + tile[wl * TAIL_PITCH + q];
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
                }
            }
            alive = alive && (m > thrC) && (h > thrC);
            if (__ballot(alive) == 0ull)
            {
                break;
            }
        }
        if (alive)
        {
            const int idx = atomicAdd(a.counts + frame, 1);
            if (idx < a.maxHits)
            {
                const int lvl = int(mine.x >> 24);
                const int n = int(mine.x & 0xffffffu);
                const int nWinR = a.levels[lvl].nWinR;
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = n / nWinR;
                hit.r = n - hit.c * nWinR;
                hit.score = h;
                a.hits[int64_t(frame) * a.maxHits + idx] = hit;
            }
        }
    }
}

// ------------------------------------------------------------------------
// k_cascade_tail_rank: queue entries without leaf codes (beyond codeCap per frame) when the cascade runs on rank cells —
// k_cascade_tail3's job without the float pyramid.  A correctness path for frames with thousands of tail windows (a very
// low cascThr), not a fast one: one wave per entry, lanes = 64 consecutive trees, every lane gathers its tree's cells
// straight from the rank pyramid in global memory; the leaves are added in tree order by the 16-lane row chains of the
// sparse tile stages (row_chain, rows in sequence), the window dies at the first prefix <= cascThr (evaluate(),
// acfDetect1.cpp:123-138: same additions, same order).
// rankNodes: per tree {off[k] = (z << 24) | (c << 12) | r of node k, thr[k] = rank index bits, hs[4]}.
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_cascade_tail_rank(TileArgs a, const TreeNode* __restrict__ rankNodes)
{
    const int lane = threadIdx.x;
    const int frame = blockIdx.x % a.nFrames;
    const int cnt = min(a.qcount[frame], a.qcap);
    const int tEnd = a.g.b[4], nT = a.nTrees - tEnd;
    const float thrC = a.cascThr;
    for (;;)
    {
        int i0 = 0;
        if (lane == 0)
        {
            i0 = atomicAdd(a.qhead + frame, 1);
        }
        i0 = __builtin_amdgcn_readfirstlane(__shfl(i0, 0));
        if (i0 >= cnt)
        {
            return;
        }
        const uint2 e = a.q[int64_t(frame) * a.qcap + i0];
        const int lvl = int(e.x >> 24), n = int(e.x & 0xffffffu);
        const CascLevel L = a.levels[lvl];
        const int c = n / L.nWinR, r = n - c * L.nWinR;
        const uint16_t* __restrict__ chn = a.pyrR + int64_t(frame) * a.pyrR_fs + L.offR + int64_t(c * a.g.step) * L.pitchR + r * a.g.step;
        const uint32_t area = uint32_t(L.pitchR) * uint32_t(L.wP);
        float h = __uint_as_float(e.y);
        bool alive = true;
        for (int tb = 0; tb < nT && alive; tb += 64) // wave-uniform
        {
            const bool act = tb + lane < nT;
            const TreeNode nd = rankNodes[tEnd + min(tb + lane, nT - 1)];
            uint32_t f[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const uint32_t zcr = nd.off[k];
                f[k] = chn[(zcr >> 24) * area + ((zcr >> 12) & 0xfffu) * uint32_t(L.pitchR) + (zcr & 0xfffu)];
            }
            const bool lt0 = f[0] < __float_as_uint(nd.thr[0]);
            const bool lt1 = (lt0 ? f[1] : f[2]) < __float_as_uint(lt0 ? nd.thr[1] : nd.thr[2]);
            float leaf = lt0 ? (lt1 ? nd.hs[0] : nd.hs[1]) : (lt1 ? nd.hs[2] : nd.hs[3]);
            leaf = act ? leaf : 0.f; // h never is -0.0f: h + 0.0f == h bit for bit
            float acc = h, mm = h;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                float sa = k == 0 ? h : dpp_row_bcast15(acc);
                float sm = k == 0 ? h : dpp_row_bcast15(mm);
                row_chain(leaf, sa, sm);
                const bool mine = (lane >> 4) == k;
                acc = mine ? sa : acc;
                mm = mine ? sm : mm;
            }
            h = __shfl(acc, 63);
            const float mAll = __shfl(mm, 63);
            alive = (mAll > thrC) && (h > thrC);
        }
        if (alive && lane == 0)
        {
            const int idx = atomicAdd(a.counts + frame, 1);
            if (idx < a.maxHits)
            {
                acf_hip_hit hit;
                hit.scale = lvl;
                hit.c = c;
                hit.r = r;
                hit.score = h;
                a.hits[int64_t(frame) * a.maxHits + idx] = hit;
            }
        }
    }
}

// ------------------------------------------------------------------------
// k_tail_scan: the ordered part of the tail [tEnd, nTrees).  Which LEAF a tree selects does not depend on the running
// score — only the early exit does (acfDetect1.cpp:123-138) — so the tile kernels' stage E writes one byte per tail tree
// of every window that reaches the tail (4 * (leaf index - 3)), and this kernel does what is sequential: lanes = windows,
// h = h + hs[leaf] strictly in tree order, 16 code bytes per 16-byte load, the leaf values of tree t read from an LDS
// table at [t][code] (all lanes of a wave hit the same 16 bytes); a lane dies at the first prefix <= cascThr.  Scores are
// bit-identical to evaluate()'s.  (A stand-alone code kernel that re-fetched each window's 16 KB footprint from HBM —
// trees in registers, windows streamed through LDS — measured 5.5 us per 1080p frame, bound by the 80-byte column runs of a
// footprint; inside the tile the features are already in LDS.)
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tail_scan(TileArgs a)
{
    extern __shared__ float lds[]; // [nT][4] leaf values of the tail trees
    const int frame = blockIdx.x % a.nFrames, chunk = blockIdx.x / a.nFrames;
    const int cnt = min(a.qcount[frame], a.qcap);
    const int cntC = min(cnt, a.codeCap);
    if (chunk == 0 && threadIdx.x == 0)
    {
        a.qhead[frame] = cntC; // k_cascade_tail3 (launched after this kernel) starts at the first entry without codes
    }
    if (chunk * 256 >= cntC)
    {
        return;
    }
    const int tEnd = a.g.b[4], nT = a.nTrees - tEnd;
    for (int t = threadIdx.x; t < nT; t += 256)
    {
        const float* hs = a.tailNodes[tEnd + t].hs;
        *reinterpret_cast<float4*>(lds + 4 * t) = make_float4(hs[0], hs[1], hs[2], hs[3]);
    }
    __syncthreads();
    const int i = chunk * 256 + int(threadIdx.x);
    bool alive = i < cntC;
    const int ic = min(i, cntC - 1);
    const uint2 e = a.q[int64_t(frame) * a.qcap + ic];
    const uint8_t* __restrict__ cp = a.tailCodes + (int64_t(frame) * a.codeCap + ic) * a.codePitch;
    const float thrC = a.cascThr;
    float h = __uint_as_float(e.y);
    float m = h; // running minimum of the prefix scores
    const char* leafB = reinterpret_cast<const char*>(lds);
    // 64 trees (four 16-byte code loads) per step, two steps requested ahead: a lane's codes are its own cache lines, so
    // every load is a full memory round trip and only distance hides it.  The three register sets swap roles in an
    // unrolled loop: copying a set would wait for the loads that fill it.
    int tb = 0;
    uint4 w0[4], w1[4], w2[4];
#define TS_LOAD(W, T0)                                                                        \
    _Pragma("unroll") for (int k = 0; k < 4; k++)                                             \
    {                                                                                         \
        W[k] = *reinterpret_cast<const uint4*>(cp + min((T0) + 16 * k, a.codePitch - 16));    \
    }
#define TS_STEP(W, T0, NQ)                                                                    \
    {                                                                                         \
        const char* lb = leafB + (T0) * 16;                                                   \
        _Pragma("unroll") for (int q = 0; q < (NQ); q++)                                      \
        {                                                                                     \
            const uint4 x = W[q >> 4];                                                        \
            const uint32_t cw = ((q >> 2) & 3) == 0 ? x.x : (((q >> 2) & 3) == 1 ? x.y : (((q >> 2) & 3) == 2 ? x.z : x.w)); \
            const uint32_t off = (cw >> (8 * (q & 3))) & 0xffu;                               \
            h = h + *reinterpret_cast<const float*>(lb + q * 16 + off);                       \
            asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));                                   \
        }                                                                                     \
    }
#define TS_ROUND(CUR, FAR)                                                                    \
    if (tb + 64 <= nT && !done)                                                               \
    {                                                                                         \
        TS_LOAD(FAR, tb + 128);                                                               \
        TS_STEP(CUR, tb, 64);                                                                 \
        alive = alive && (m > thrC) && (h > thrC);                                            \
        done = __ballot(alive) == 0ull;                                                       \
        tb += done ? 0 : 64;                                                                  \
    }
    TS_LOAD(w0, 0);
    TS_LOAD(w1, 64);
    bool done = false;
    while (tb + 64 <= nT && !done)
    {
        TS_ROUND(w0, w2);
        TS_ROUND(w1, w0);
        TS_ROUND(w2, w1);
    }
    // the set holding the codes of [tb, tb + 64): rounds completed mod 3
    {
        const int rr = (tb >> 6) % 3;
        if (rr == 1)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                w0[k] = w1[k];
            }
        }
        else if (rr == 2)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                w0[k] = w2[k];
            }
        }
    }
#undef TS_ROUND
    if (done)
    {
        tb = nT; // every lane of the wave is rejected: nothing left to add
    }
    if (tb + 64 > nT && tb < nT) // fewer than 64 trees left: w0 holds their codes
    {
        const int rem = nT - tb;
        const char* lb = leafB + tb * 16;
#pragma unroll
        for (int q = 0; q < 64; q++)
        {
            if (q >= rem)
            {
                break;
            }
            const uint4 x = w0[q >> 4];
            const uint32_t cw = ((q >> 2) & 3) == 0 ? x.x : (((q >> 2) & 3) == 1 ? x.y : (((q >> 2) & 3) == 2 ? x.z : x.w));
            const uint32_t off = (cw >> (8 * (q & 3))) & 0xffu;
            h = h + *reinterpret_cast<const float*>(lb + q * 16 + off);
            asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(h));
        }
    }
#undef TS_LOAD
#undef TS_STEP
    alive = alive && (m > thrC) && (h > thrC);
    const unsigned long long mask = __ballot(alive);
    if (mask)
    {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0)
        {
            base = atomicAdd(a.counts + frame, __popcll(mask));
        }
        base = __shfl(base, 0);
        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (alive && idx < a.maxHits)
        {
            const int lvl = int(e.x >> 24);
            const int n = int(e.x & 0xffffffu);
            const int nWinR = a.levels[lvl].nWinR;
            acf_hip_hit hit;
            hit.scale = lvl;
            hit.c = n / nWinR;
            hit.r = n - hit.c * nWinR;
            hit.score = h;
            a.hits[int64_t(frame) * a.maxHits + idx] = hit;
        }
    }
}

// Detector::evaluate(const MatP&, ...) (acfDetect1.cpp:337-342): the score of the single window at (0, 0), trees added in
// order until h <= cascThr (the reference sets cascThr = 0 for this call) — evaluate(), :113-138, with getChild (:100-107) or
// the child-pointer walk (:146-155).  One thread: this is a probe, not a hot path.
__global__ void k_evaluate_window(const float* __restrict__ chns, int hP, int wP, int mH, int mW, const uint32_t* __restrict__ fids,
    const float* __restrict__ thrs, const float* __restrict__ hs, const uint32_t* __restrict__ child, int nTrees, int nTreeNodes, int depth,
    float cascThr, float* __restrict__ score)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
    {
        return;
    }
    const int area = hP * wP;
    float h = 0.f;
    for (int t = 0; t < nTrees; t++)
    {
        const uint32_t offset = uint32_t(t) * uint32_t(nTreeNodes);
        uint32_t k = offset, k0 = depth == 0 ? k : 0u;
        if (depth > 0)
        {
            for (int i = 0; i < depth; i++)
            {
                const uint32_t f = fids[k];
                const uint32_t z = f / uint32_t(mW * mH), cc = (f / uint32_t(mH)) % uint32_t(mW), rr = f % uint32_t(mH); // cids[], :390-406
                const float ftr = chns[z * uint32_t(area) + cc * uint32_t(hP) + rr];
                k = (ftr < thrs[k]) ? 1u : 2u;
                k0 = k += k0 * 2u;
                k += offset;
            }
        }
        else
        {
            while (child[k])
            {
                const uint32_t f = fids[k];
                const uint32_t z = f / uint32_t(mW * mH), cc = (f / uint32_t(mH)) % uint32_t(mW), rr = f % uint32_t(mH);
                const float ftr = chns[z * uint32_t(area) + cc * uint32_t(hP) + rr];
                k = (ftr < thrs[k]) ? 1u : 0u;
                k0 = k = child[k0] - k + offset;
            }
        }
        h += hs[k];
        if (h <= cascThr)
        {
            break;
        }
    }
    *score = h;
}

// Sort each frame's hits into the reference's order (level, then c, then r:
// ACF.cpp:326-329, acfDetect1.cpp:86-96) by rank counting, and map them to
// image boxes (ACF.cpp:302-312).  Hit lists are small (<= maxHits), the keys
// are unique, so every hit's rank is the number of hits with a smaller key.
struct BoxLevel
{
    double shw_h, shw_w;
    int32_t bw, bh; // cvRound(modelDs / scale), precomputed on the host (ACF.cpp:304)
};

constexpr int SM_BLOCKS = 32; // workgroups per frame (256 threads each); blocks without items leave at once
constexpr int SM_ITEMS = 8;   // hits per thread and pass
constexpr int SM_CHUNK = 2048;

template <int ITEMS>
__device__ __forceinline__ void sort_map_body(const acf_hip_hit* __restrict__ H, int n, int frame, int maxHits, const BoxLevel* __restrict__ bl, int stride,
    int shift_h, int shift_w, acf_hip_hit* __restrict__ sortedHits, acf_hip_detection* __restrict__ dets, long long* keys)
{
    const int nThreads = SM_BLOCKS * 256;
    for (int i0 = 0; i0 < n; i0 += nThreads * ITEMS)
    {
        const int first = i0 + (blockIdx.x * 256 + threadIdx.x) * ITEMS;
        if (i0 + blockIdx.x * 256 * ITEMS >= n) // block-uniform: nothing for this block in this pass (nor in later ones)
        {
            return;
        }
        acf_hip_hit me[ITEMS];
        long long key[ITEMS];
        int rank[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
        {
            me[q] = H[min(first + q, n - 1)];
            key[q] = (((long long)me[q].scale) << 40) | (((long long)me[q].c) << 20) | (long long)me[q].r;
            rank[q] = 0;
        }
        for (int c0 = 0; c0 < n; c0 += SM_CHUNK)
        {
            const int m = min(SM_CHUNK, n - c0);
            __syncthreads();
            for (int j = threadIdx.x; j < m; j += 256)
            {
                const acf_hip_hit o = H[c0 + j];
                keys[j] = (((long long)o.scale) << 40) | (((long long)o.c) << 20) | (long long)o.r;
            }
            __syncthreads();
            for (int j = 0; j < m; j++)
            {
                const long long ko = keys[j];
#pragma unroll
                for (int q = 0; q < ITEMS; q++)
                {
                    rank[q] += ko < key[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
        {
            if (first + q < n)
            {
                sortedHits[int64_t(frame) * maxHits + rank[q]] = me[q];
                const BoxLevel b = bl[me[q].scale];
                acf_hip_detection d;
                // roi = ({c*stride, r*stride}); x = int(double(x + shift)/scaleshw) (truncation)
                d.x = (int)((double)(me[q].c * stride + shift_w) / b.shw_w);
                d.y = (int)((double)(me[q].r * stride + shift_h) / b.shw_h);
                d.w = b.bw;
                d.h = b.bh;
                d.score = me[q].score;
                d.scale = me[q].scale;
                dets[int64_t(frame) * maxHits + rank[q]] = d;
            }
        }
    }
}

// stride < shrink (LDCF's default: stride 4 on cells of 8 pixels): acfDetect1 places window (r, c) at cell offset
// (r * stride / shrink, c * stride / shrink) — integer division (acfDetect1.cpp:88-96) —, so shrink / stride consecutive rows
// and columns of windows read the SAME cells and get the same score.  The cascade then runs once per distinct offset (CascLevel
// carries the distinct grid, CascArgs::stride = shrink) and this kernel writes every window of a surviving offset: hit (r', c')
// of the distinct grid -> windows r' q .. r' q + q - 1 (< nWinR), c' q .. (< nWinC), q = shrink / stride, all with its score.
// One workgroup per frame; k_sort_map orders the result by (level, c, r) whatever order it was written in.
__global__ void __launch_bounds__(256) k_expand_hits(const acf_hip_hit* __restrict__ in, int32_t* __restrict__ counts, acf_hip_hit* __restrict__ out, int maxHits,
    const int2* __restrict__ realWin, int q)
{
    __shared__ int s_total;
    const int frame = blockIdx.x;
    const int n = counts[frame], m = min(n, maxHits);
    if (threadIdx.x == 0)
    {
        s_total = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256)
    {
        const acf_hip_hit hd = in[int64_t(frame) * maxHits + i];
        const int2 rw = realWin[hd.scale];
        const int r0 = hd.r * q, c0 = hd.c * q;
        const int nr = min(q, rw.x - r0), nc = min(q, rw.y - c0);
        const int base = atomicAdd(&s_total, nr * nc);
        for (int dc = 0; dc < nc; dc++)
        {
            for (int dr = 0; dr < nr; dr++)
            {
                const int idx = base + dc * nr + dr;
                if (idx < maxHits)
                {
                    acf_hip_hit h;
                    h.scale = hd.scale;
                    h.c = c0 + dc;
                    h.r = r0 + dr;
                    h.score = hd.score;
                    out[int64_t(frame) * maxHits + idx] = h;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        counts[frame] = n > maxHits ? max(n, s_total) : s_total; // (a count above maxHits is the caller's overflow signal either way)
    }
}

__global__ void __launch_bounds__(256) k_sort_map(const acf_hip_hit* __restrict__ hits, const int32_t* __restrict__ counts,
    int maxHits, const BoxLevel* __restrict__ bl, int stride, int shift_h, int shift_w,
    acf_hip_hit* __restrict__ sortedHits, acf_hip_detection* __restrict__ dets)
{
    // Rank sort on the 64-bit key (scale, c, r) — the order acfDetect1's loops emit (ACF.cpp:283-300).  The keys are
    // staged through LDS SM_CHUNK at a time.  A frame whose hits fit the grid (the usual few hundred) gives every hit
    // its own thread; beyond that a thread ranks SM_ITEMS hits per key read, so a frame near capacity (65,536 hits: a
    // low cascThr) costs 4e9 LDS compares spread over 8192 threads instead of 4e9 global reads on 1024 (round 1).
    __shared__ long long keys[SM_CHUNK];
    const int frame = blockIdx.y;
    const int n = min(counts[frame], maxHits);
    const acf_hip_hit* H = hits + int64_t(frame) * maxHits;
    if (n <= SM_BLOCKS * 256)
    {
        sort_map_body<1>(H, n, frame, maxHits, bl, stride, shift_h, shift_w, sortedHits, dets, keys);
    }
    else
    {
        sort_map_body<SM_ITEMS>(H, n, frame, maxHits, bl, stride, shift_h, shift_w, sortedHits, dets, keys);
    }
}

// ------------------------------------------------------------------------
// bbNms (max / maxg) + ObjectDetector::prune on the device: one workgroup per frame, before the records are exported
// or gathered (bbNms.cpp:111-192, 229-304; ObjectDetector.cpp:28-44).
//   1. scores below thr are dropped (bbNms.cpp:276-279);
//   2. the rest is ordered by score, descending.  The reference uses std::sort (util/ordered.h:23-31), whose order
//      among EQUAL scores is unspecified; here ties keep their input order (scale, column, row) — one of its valid
//      outcomes, and deterministic.  Bitonic sort of {orderable f64 key, index} in LDS;
//   3. for i in order: (maxg: only if i itself survived) every later j with area-overlap(i, j) > overlap is suppressed.
//      overlap = iw*ih / (union | min area), computed in f64 from int products exactly as :158-165.  The outer loop is
//      the reference's sequential dependency; the inner loop runs across the workgroup;
//   4. survivors are emitted in order; prune keeps cutoff of them (ObjectDetector.cpp:30-42: up to maxCount, and one
//      past the first score below scores[0] * ratio).
// Output: indices into the frame's input in output order (+ the gathered records for the pipeline form).
// ------------------------------------------------------------------------
constexpr int NMS_CAP = 2048; // 58 KB of LDS: a block can start next to a resident cascade tile workgroup (81 KB)

struct NmsArgs
{
    // pipeline form: detections [frame][maxHits] (f32 scores) and their counts
    const acf_hip_detection* dets;
    const int32_t* counts;
    int32_t maxHits;
    // op form (dets == nullptr): one list of n boxes {x, y, w, h} and f64 scores
    const int32_t* boxes;
    const double* scores;
    int32_t nOp;
    int32_t greedy, ovrUnion, doPrune, maxCount;
    double overlap, thr, pruneRatio;
    int32_t* keep;              // [frame][NMS_CAP]
    int32_t* nKeep;             // [frame]: survivors, or -1: more than NMS_CAP inputs
    acf_hip_detection* outDets; // pipeline form: [frame][maxHits]
    int32_t* outCounts;
};

__device__ __forceinline__ unsigned long long nms_key(double v) // monotonic: larger score -> larger key
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void __launch_bounds__(1024) k_nms(NmsArgs a)
{
    extern __shared__ unsigned long long nms_lds[];
    unsigned long long* key = nms_lds;                                // [NMS_CAP]
    int4* box = reinterpret_cast<int4*>(key + NMS_CAP);               // [NMS_CAP] {xs, ys, xe, ye} in sorted order
    uint32_t* idx = reinterpret_cast<uint32_t*>(box + NMS_CAP);       // [NMS_CAP]
    uint8_t* kp = reinterpret_cast<uint8_t*>(idx + NMS_CAP);          // [NMS_CAP]
    __shared__ int s_drop, s_wave[16], s_cut;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const bool pipe = a.dets != nullptr;
    const int n = pipe ? min(a.counts[frame], a.maxHits) : a.nOp;
    const acf_hip_detection* __restrict__ D = pipe ? a.dets + int64_t(frame) * a.maxHits : nullptr;
    int32_t* __restrict__ keep = a.keep + int64_t(frame) * NMS_CAP;
    if (n > NMS_CAP)
    {
        if (tid == 0)
        {
            a.nKeep[frame] = -1;
            if (pipe)
            {
                a.outCounts[frame] = -1;
            }
        }
        return;
    }
    int P = 1;
    while (P < n)
    {
        P <<= 1;
    }
    // As many threads as the sort has compare-exchange pairs (at least a wave): the waves beyond them leave before the first
    // barrier — a barrier among 4 waves costs a quarter of one among 16, and the kernel is a sequence of ~100 barriers
    // (one frame of 360 raw detections: 86 -> 35 us)
    const int T = min(1024, max(64, P >> 1));
    if (tid >= T)
    {
        return;
    }
    if (tid == 0)
    {
        s_drop = 0;
        s_cut = 0x7fffffff;
    }
    if (tid < 16)
    {
        s_wave[tid] = 0;
    }
    __syncthreads();
    // ---- 1. keys (dropped and padding entries sort last)
    for (int i = tid; i < P; i += T)
    {
        unsigned long long k = 0ull;
        uint32_t ix = 0xffffffffu;
        if (i < n)
        {
            const double sc = pipe ? double(D[i].score) : a.scores[i];
            if (sc < a.thr)
            {
                atomicAdd(&s_drop, 1);
            }
            else
            {
                k = nms_key(sc);
                // keys of real entries are never 0: the smallest is nms_key(-inf) > 0... NaN payloads aside; keep 0 for padding
                k = k ? k : 1ull;
                ix = uint32_t(i);
            }
        }
        key[i] = k;
        idx[i] = ix;
    }
    __syncthreads();
    const int m = n - s_drop;
    // ---- 2. bitonic sort, "before" = larger key, then smaller index
    for (int k2 = 2; k2 <= P; k2 <<= 1)
    {
        for (int j = k2 >> 1; j > 0; j >>= 1)
        {
            for (int t = tid; t < (P >> 1); t += T)
            {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k2) == 0; // ascending block: "before" elements first
                const unsigned long long ka = key[lo], kb = key[hi];
                const uint32_t ia = idx[lo], ib = idx[hi];
                const bool aFirst = ka > kb || (ka == kb && ia < ib);
                if (aFirst != up)
                {
                    key[lo] = kb;
                    key[hi] = ka;
                    idx[lo] = ib;
                    idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    // ---- boxes in sorted order
    for (int i = tid; i < m; i += T)
    {
        const uint32_t q = idx[i];
        int x, y, w, h;
        if (pipe)
        {
            x = D[q].x, y = D[q].y, w = D[q].w, h = D[q].h;
        }
        else
        {
            x = a.boxes[4 * q], y = a.boxes[4 * q + 1], w = a.boxes[4 * q + 2], h = a.boxes[4 * q + 3];
        }
        box[i] = make_int4(x, y, x + w, y + h);
        kp[i] = 1;
    }
    __syncthreads();
    // ---- 3. suppression
    for (int i = 0; i + 1 < m; i++)
    {
        if (a.greedy && !kp[i]) // uniform: every thread reads the same byte
        {
            continue;
        }
        const int4 bi = box[i];
        const int asI = (bi.z - bi.x) * (bi.w - bi.y);
        for (int j = i + 1 + tid; j < m; j += T)
        {
            if (!kp[j])
            {
                continue;
            }
            const int4 bj = box[j];
            const int iw = min(bi.z, bj.z) - max(bi.x, bj.x);
            const int ih = min(bi.w, bj.w) - max(bi.y, bj.y);
            if (iw <= 0 || ih <= 0)
            {
                continue;
            }
            const int asJ = (bj.z - bj.x) * (bj.w - bj.y);
            double o = double(iw * ih);
            const double u = a.ovrUnion ? (double(asI + asJ) - o) : double(min(asI, asJ));
            o /= u;
            if (o > a.overlap)
            {
                kp[j] = 0;
            }
        }
        __syncthreads();
    }
    // ---- 4. survivors in order: exclusive scan over 4 entries per thread
    const int lane = tid & 63, wv = tid >> 6;
    int c4[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int i = 4 * tid + q;
        c4[q] = (i < m && kp[i]) ? 1 : 0;
        tot += c4[q];
    }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const int v = __shfl_up(incl, d);
        incl += (lane >= d) ? v : 0;
    }
    if (lane == 63)
    {
        s_wave[wv] = incl;
    }
    __syncthreads();
    int base = incl - tot, total = 0;
    for (int q = 0; q < 16; q++)
    {
        base += (q < wv) ? s_wave[q] : 0;
        total += s_wave[q];
    }
    // prune (ObjectDetector.cpp:28-44) on the ordered survivors: needs their scores -> stage the output positions first
    uint32_t* outIdx = reinterpret_cast<uint32_t*>(key); // the keys are dead: reuse as [total] original indices in output order
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        if (c4[q])
        {
            outIdx[base] = idx[4 * tid + q];
            base++;
        }
    }
    __syncthreads();
    int finalCount = total;
    if (a.doPrune && total > 1)
    {
        const int L = min(a.maxCount, total);
        const uint32_t q0 = outIdx[0];
        const double s0 = pipe ? double(D[q0].score) : a.scores[q0];
        for (int i = 1 + tid; i < L; i += T)
        {
            const uint32_t qi = outIdx[i];
            const double si = pipe ? double(D[qi].score) : a.scores[qi];
            if (si < s0 * a.pruneRatio)
            {
                atomicMin(&s_cut, i);
            }
        }
        __syncthreads();
        finalCount = L < 2 ? 1 : (s_cut < L ? s_cut + 1 : L);
    }
    for (int i = tid; i < finalCount; i += T)
    {
        const uint32_t q = outIdx[i];
        keep[i] = int32_t(q);
        if (pipe)
        {
            a.outDets[int64_t(frame) * a.maxHits + i] = D[q];
        }
    }
    if (tid == 0)
    {
        a.nKeep[frame] = finalCount;
        if (pipe)
        {
            // a truncated input (more hits than max_hits) keeps reporting the overflow through the count
            a.outCounts[frame] = a.counts[frame] > a.maxHits ? a.counts[frame] : finalCount;
        }
    }
}

__global__ void __launch_bounds__(256) k_export(const acf_hip_detection* __restrict__ dets, const int32_t* __restrict__ counts,
    int maxHits, int cap, int32_t* __restrict__ dst)
{
    const int frame = blockIdx.y;
    // more hits than max_hits: which ones were kept depends on the order of atomics, so no record is exported — the count
    // (> max_hits) tells the consumer; a frame over the device NMS capacity carries count -1
    const int n = counts[frame] > maxHits ? 0 : min(min(counts[frame], maxHits), cap);
    int32_t* D = dst + int64_t(frame) * (1 + 6 * int64_t(cap));
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        D[0] = counts[frame];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x)
    {
        int32_t* R = D + 1 + 6 * int64_t(i);
        if (i < n)
        {
            const acf_hip_detection d = dets[int64_t(frame) * maxHits + i];
            R[0] = d.x;
            R[1] = d.y;
            R[2] = d.w;
            R[3] = d.h;
            R[4] = __float_as_int(d.score);
            R[5] = d.scale;
        }
        else
        {
            R[0] = R[1] = R[2] = R[3] = R[4] = R[5] = 0;
        }
    }
}

} // namespace acfhip
