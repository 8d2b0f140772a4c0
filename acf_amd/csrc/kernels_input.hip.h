// kernels_input.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// rgbConvert (LUV / gray / HSV), the packed 8-bit ingest, the apps' resize to a minimum object width, and the plain convTri1 (k_smooth_tri1: the fallback of the vector smoothing).
#pragma once

namespace acfhip
{

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
// x86: the table of the CPU whose _mm_rcp_ps the SSE body's one reciprocal (:161) reproduces (option "arith"), or null: 1 / x.
template <bool VEC>
__device__ __forceinline__ void luv_px(float r, float g, float b, const float* __restrict__ lTable, const LuvConsts& k, float& L, float& U, float& V,
    const uint32_t* __restrict__ x86 = nullptr)
{
    if (VEC)
    {
        const float x = (r * k.mr[0] + g * k.mg[0]) + b * k.mb[0];
        const float y = (r * k.mr[1] + g * k.mg[1]) + b * k.mb[1];
        const float z = (r * k.mr[2] + g * k.mg[2]) + b * k.mb[2];
        const float den = x + (1e-35f + (15.0f * y + 3.0f * z));
        const float zz = x86 ? x86_rcp(den, x86) : 1.0f / den;
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
    const float* __restrict__ lTable, LuvConsts k, int n, int64_t in_fs, int64_t out_fs, const uint32_t* __restrict__ x86)
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
    luv_px<VEC>(r, g, b, lTable, k, L, U, V, x86);
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
    const uint32_t* x86; // option "arith": the CPU tables for rgb2luv_sse's reciprocal (null: exact)
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
                luv_px<MODE == IG_LUV_VEC>(r, g, b, a.lTable, a.k, o[0][j], o[1][j], o[2][j], a.x86);
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

} // namespace acfhip
