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
struct SmoothJob
{
    int32_t h, w, nplanes, out_cs; // out_cs: destination column stride (hP)
    int64_t in_off, out_off;       // float offsets inside a frame's source / destination buffer
    int64_t in_ps, out_ps;         // plane strides
};

template <int R, bool ALIASED>
__global__ void __launch_bounds__(1024) k_smooth_tri1(const float* __restrict__ in, float* __restrict__ out,
    const SmoothJob* __restrict__ jobs, int64_t in_fs, int64_t out_fs, float p, int ldsStride)
{
    extern __shared__ float lds[]; // 2 * ldsStride floats
    const SmoothJob job = jobs[blockIdx.y];
    if ((int)blockIdx.x >= job.nplanes)
    {
        return;
    }
    const int h = job.h, w = job.w;
    const float* I = in + int64_t(blockIdx.z) * in_fs + job.in_off + int64_t(blockIdx.x) * job.in_ps;
    float* O = out + int64_t(blockIdx.z) * out_fs + job.out_off + int64_t(blockIdx.x) * job.out_ps;
    const int tid = threadIdx.x, nt = blockDim.x;
    const float nrm = 1.0f / ((p + 2) * (p + 2));
    const float p1 = 1 + p;

    float c[4][R], nx[4][R], prev[R];
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
#pragma unroll
        for (int k = 0; k < R; k++)
        {
            const int y = tid + k * nt;
            c[j][k] = (j < w && y < h) ? I[int64_t(j) * h + y] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < R; k++)
    {
        prev[k] = c[0][k]; // Il = Im at i == 0 (:503-507)
    }
    int buf = 0;
    for (int i0 = 0; i0 < w; i0 += 4)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                const int y = tid + k * nt;
                const int col = i0 + 4 + j;
                nx[j][k] = (col < w && y < h) ? I[int64_t(col) * h + y] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int i = i0 + j;
            if (i < w) // uniform across the workgroup
            {
                float* Tb = lds + buf * ldsStride;
                float T[R];
#pragma unroll
                for (int k = 0; k < R; k++)
                {
                    const int y = tid + k * nt;
                    const float Im = c[j][k];
                    const float Irn = (j < 3) ? c[(j + 1) & 3][k] : nx[0][k];
                    const float Ir = (i < w - 1) ? Irn : Im;
                    const float Il = ALIASED ? prev[k] : ((i > 0) ? ((j > 0) ? c[(j + 3) & 3][k] : prev[k]) : Im);
                    T[k] = nrm * (Il + p * Im + Ir);
                    if (y < h)
                    {
                        Tb[y] = T[k];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < R; k++)
                {
                    const int y = tid + k * nt;
                    if (y < h)
                    {
                        float o;
                        if (y == 0)
                        {
                            o = p1 * T[k] + Tb[1];
                        }
                        else if (y == h - 1)
                        {
                            o = Tb[y - 1] + p1 * T[k];
                        }
                        else
                        {
                            o = Tb[y - 1] + p * T[k] + Tb[y + 1];
                        }
                        O[int64_t(i) * job.out_cs + y] = o;
                        if (ALIASED)
                        {
                            prev[k] = o;
                        }
                    }
                }
                buf ^= 1;
            }
        }
        if (!ALIASED)
        {
            // non-aliased: Il of the next chunk's first column is this chunk's last input column
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                prev[k] = c[3][k];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
#pragma unroll
            for (int k = 0; k < R; k++)
            {
                c[j][k] = nx[j][k];
            }
        }
    }
}

// cv::copyMakeBorder(BORDER_REFLECT) of the interior already written by the
// smoothing kernel (chnsPyramid.cpp:410-424): fills only the border cells.
struct PadJob
{
    int32_t hC, wC, hP, wP, py, px, nplanes, pad_;
    int64_t off; // level offset in the fused pyramid
};

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    while (i < 0 || i >= n)
    {
        i = (i < 0) ? (-i - 1) : (2 * n - 1 - i);
    }
    return i;
}

__global__ void __launch_bounds__(256) k_pad_reflect(float* __restrict__ pyr, const PadJob* __restrict__ jobs, int64_t fs)
{
    const PadJob j = jobs[blockIdx.y];
    const int64_t cells = int64_t(j.hP) * j.wP;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= cells * j.nplanes)
    {
        return;
    }
    const int c = int(e / cells);
    const int rem = int(e - int64_t(c) * cells);
    const int x = rem / j.hP, y = rem - x * j.hP;
    const int sx = x - j.px, sy = y - j.py;
    if (sx >= 0 && sx < j.wC && sy >= 0 && sy < j.hC)
    {
        return; // interior
    }
    float* P = pyr + int64_t(blockIdx.z) * fs + j.off + int64_t(c) * cells;
    const int rx = reflect_idx(sx, j.wC) + j.px, ry = reflect_idx(sy, j.hC) + j.py;
    P[int64_t(x) * j.hP + y] = P[int64_t(rx) * j.hP + ry];
}

// ------------------------------------------------------------------------
// gradMag, d == 1 (toolbox/gradientMex.cpp:17-87,168-251).  acosT points at
// the table's centre (index 0).
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_grad_mag(const float* __restrict__ in, float* __restrict__ M, float* __restrict__ O,
    const float* __restrict__ acosT, int h, int w, int full, int64_t in_fs, int64_t out_fs)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = blockIdx.y;
    if (y >= h)
    {
        return;
    }
    const float* I = in + int64_t(blockIdx.z) * in_fs + int64_t(x) * h;
    // grad1 :22-53
    const float* Ip = I - h;
    const float* In = I + h;
    float r = .5f;
    if (x == 0)
    {
        r = 1;
        Ip += h;
    }
    else if (x == w - 1)
    {
        r = 1;
        In -= h;
    }
    const float gx = (In[y] - Ip[y]) * r;
    // :58
    float gy;
    if (y == 0)
    {
        gy = (I[1] - I[0]) * 1;
    }
    else if (y == h - 1)
    {
        gy = (I[h - 1] - I[h - 2]) * 1;
    }
    else
    {
        gy = (I[y + 1] - I[y - 1]) * .5f;
    }
    const float m2 = gx * gx + gy * gy;
    float m = 1.0f / sqrtf(m2);
    m = m < 1e10f ? m : 1e10f;
    const int64_t o = int64_t(blockIdx.z) * out_fs + int64_t(x) * h + y;
    M[o] = 1.0f / m;
    float g = (gx * m) * 10000.0f;
    g = __int_as_float(__float_as_int(g) ^ (__float_as_int(gy) & 0x80000000));
    g = g < 10009.0f ? g : 10009.0f;
    g = g > -10009.0f ? g : -10009.0f;
    float ov = acosT[(int)g];
    if (full)
    {
        ov += (gy < 0) * 3.14159265f;
    }
    O[o] = ov;
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
    // body: r < i <= w - r, loads independent of the recurrence (4 columns in flight)
    for (; i + 3 <= w - r; i += 4)
    {
        float a[4], b[4], c[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            a[j] = I[int64_t(i + j - 1 - r) * h];
            b[j] = I[int64_t(i + j - 1 + r) * h];
            c[j] = I[int64_t(i + j - 1) * h];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            T += a[j] + b[j] - 2 * c[j];
            U += nrm * T;
            Uc[int64_t(i + j) * h] = U;
        }
    }
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
    float normConst, rq; // rq = (1/S)/(1+1e-6) then /S in the y pass (imResampleMex.cpp:145-157,316)
    float rq_y;
};

template <int S>
__global__ void __launch_bounds__(256) k_chns(ChnsArgs a)
{
    const int hc = a.h / S, wc = a.w / S;
    const int yc = blockIdx.x * blockDim.x + threadIdx.x;
    const int xc = blockIdx.y;
    if (yc >= hc)
    {
        return;
    }
    const int64_t f = blockIdx.z;
    const int64_t cells = int64_t(hc) * wc;
    float* out = a.chns + f * a.chns_fs + int64_t(xc) * hc + yc;
    const int64_t pbase = int64_t(xc * S) * a.h + yc * S;
    int ch = 0;
    if (a.colorEnabled)
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
#pragma unroll
        for (int xx = 0; xx < S; xx++)
        {
#pragma unroll
            for (int yy = 0; yy < S; yy++)
            {
                float m = Mp[int64_t(xx) * a.h + yy];
                if (a.doNorm)
                {
                    const float s = Sp[int64_t(xx) * a.h + yy];
                    // vector body of gradMagNorm: M * rcp(S + norm); the scalar tail
                    // (last n%4 elements) divides — n%4 == 0 here since h % shrink == 0, shrink in {2,4}... see launch
                    m = m * (1.0f / (s + a.normConst));
                }
                mn[xx][yy] = m;
                ov[xx][yy] = Op[int64_t(xx) * a.h + yy];
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
        constexpr int MAXO = 12;
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
                int o0 = (int)o;
                const float od = o - (float)o0;
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
#pragma unroll
                for (int b = 0; b < MAXO; b++)
                {
                    H[b] = (b == o0) ? H[b] + m0 : H[b];
                }
#pragma unroll
                for (int b = 0; b < MAXO; b++)
                {
                    H[b] = (b == o1) ? H[b] + m1 : H[b];
                }
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
__device__ __forceinline__ float rs_C(const ResampleDesc& d, const int32_t* __restrict__ it, const float* __restrict__ ft,
    const float* __restrict__ A, int xb, int row)
{
    if (row >= d.ha)
    {
        return 0.f; // C[ha .. ha+3] = 0 (:133-137)
    }
    if (d.xmode == RS_EXACT)
    {
        const float* A0 = A + int64_t(it[d.x_src + xb]) * d.ha + row;
        float s = A0[0] + A0[d.ha];
        if (d.xk > 2)
        {
            s = s + A0[2 * int64_t(d.ha)];
        }
        if (d.xk > 3)
        {
            s = s + A0[3 * int64_t(d.ha)];
        }
        return s;
    }
    if (d.xmode == RS_DOWN)
    {
        const int s0 = it[d.x_start + xb], s1 = it[d.x_start + xb + 1];
        const float* A0 = A + int64_t(it[d.x_src + s0]) * d.ha + row;
        float s = A0[0] * ft[d.x_wt + s0];
        for (int j = 1; j < s1 - s0; j++)
        {
            s = s + A0[int64_t(j) * d.ha] * ft[d.x_wt + s0 + j];
        }
        return s;
    }
    const float* A0 = A + int64_t(it[d.x_src + xb]) * d.ha + row;
    const bool xBd = xb < d.xbd0 || xb >= d.wb - d.xbd1;
    if (xBd)
    {
        return A0[0];
    }
    const float wt = ft[d.x_wt + xb];
    const float wt1 = 1 - wt;
    return A0[0] * wt + A0[d.ha] * wt1;
}

__global__ void __launch_bounds__(256) k_resample(const float* __restrict__ src, float* __restrict__ dst,
    const ResampleDesc* __restrict__ descs, const int32_t* __restrict__ it, const float* __restrict__ ft)
{
    const ResampleDesc d = descs[blockIdx.y];
    const int64_t per = int64_t(d.hb) * d.wb;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= per * d.nplanes)
    {
        return;
    }
    const int z = int(e / per);
    const int rem = int(e - int64_t(z) * per);
    const int xb = rem / d.hb, yb = rem - xb * d.hb;
    const int ty = z < d.c1 ? 0 : (z < d.c2 ? 1 : 2);
    const float r = d.r[ty];
    const float* A = src + int64_t(blockIdx.z) * d.src_frame_stride + d.src_off + int64_t(z) * d.ha * d.wa;
    float v;
    if (d.ymode == RS_EXACT)
    {
        const int k = d.yk;
        float s = rs_C(d, it, ft, A, xb, k * yb) + rs_C(d, it, ft, A, xb, k * yb + 1);
        if (k > 2)
        {
            s = s + rs_C(d, it, ft, A, xb, k * yb + 2);
        }
        if (k > 3)
        {
            s = s + rs_C(d, it, ft, A, xb, k * yb + 3);
        }
        v = s * d.rk[ty];
    }
    else if (d.ymode == RS_DOWN)
    {
        const int s0 = it[d.y_start + yb], s1 = it[d.y_start + yb + 1];
        if (d.ybd0 <= 4)
        {
            // U(0)+U(1)(+U(2)(+U(3))) with exactly ybd0 terms, rows ya+o (:324-348)
            const int ya = it[d.y_src + s0];
            v = rs_C(d, it, ft, A, xb, ya) * (ft[d.y_wt + s0] * r);
            for (int o = 1; o < d.ybd0; o++)
            {
                v = v + rs_C(d, it, ft, A, xb, ya + o) * (ft[d.y_wt + s0 + o] * r);
            }
        }
        else
        {
            // B0 zeroed then += over this output's entries in order (:349-356)
            v = 0.f;
            for (int q = s0; q < s1; q++)
            {
                v = v + rs_C(d, it, ft, A, xb, it[d.y_src + q]) * (ft[d.y_wt + q] * r);
            }
        }
    }
    else
    {
        const int ya = it[d.y_src + yb];
        const float wy = ft[d.y_wt + yb] * r;
        if (yb < d.ybd0 || yb >= d.hb - d.ybd1)
        {
            v = rs_C(d, it, ft, A, xb, ya) * wy;
        }
        else
        {
            v = rs_C(d, it, ft, A, xb, ya) * wy + rs_C(d, it, ft, A, xb, ya + 1) * (r - wy);
        }
    }
    dst[int64_t(blockIdx.z) * d.dst_frame_stride + d.dst_off + int64_t(z) * per + rem] = v;
}

// ------------------------------------------------------------------------
// The cascade: ParallelDetectionBody::operator()/evaluate
// (toolbox/acfDetect1.cpp:84-138) for every window of every level of every
// frame in one launch.  One lane per window, lanes consecutive along r (the
// contiguous image-y axis) so each feature fetch of a wave is a contiguous
// segment of the level's channel buffer.
//
// Depth-2 fast path: the three internal nodes' channel offsets and
// thresholds and the four leaf values of tree t are wave-uniform, so they are
// fetched with scalar loads from a per-level packed table; the only vector
// memory traffic is three feature fetches per tree.  A wave leaves the tree
// loop as soon as a ballot shows that none of its windows is still alive
// (early cascade rejection); hits are compacted with a ballot prefix sum and
// one atomic per wave.
// ------------------------------------------------------------------------
struct CascLevel
{
    int32_t hP, wP, nWinR, nWinC;
    int32_t firstBlock; // first block index of this level inside one frame's grid
    int32_t nWin;
    int64_t off;        // level offset in the fused pyramid
    int64_t nodeOff;    // offset (in units of CascNode2 / uint32) of this level's node table
};

struct __attribute__((aligned(16))) CascNode2
{
    uint32_t cid[4]; // cid[3] unused
    float thr[4];    // thr[3] unused
    float hs[4];     // leaves 3..6
};

struct CascArgs
{
    const float* pyr;
    int64_t pyr_fs;
    const CascLevel* levels;
    const int32_t* blockLevel; // block -> level
    int32_t blocksPerFrame;
    int32_t nTrees, nTreeNodes, treeDepth;
    int32_t stride, shrink;
    float cascThr;
    // generic path tables
    const uint32_t* cidAll;  // [level][nTrees*nTreeNodes]
    const float* thrs;       // [nTrees*nTreeNodes]
    const float* hs;
    const uint32_t* child;
    const CascNode2* nodes2; // depth-2 packed tables
    // output
    acf_hip_hit* hits; // [frame][maxHits]
    int32_t* counts;   // [frame]
    int32_t maxHits;
};

template <int MODE> // 2: packed depth-2 path; 1: generic fixed depth; 0: child walk
__global__ void __launch_bounds__(256) k_cascade(CascArgs a)
{
    const int frame = blockIdx.y;
    const int lvl = a.blockLevel[blockIdx.x];
    const CascLevel L = a.levels[lvl];
    const int n = (blockIdx.x - L.firstBlock) * blockDim.x + threadIdx.x;
    bool alive = n < L.nWin;
    const int c = alive ? n / L.nWinR : 0;
    const int r = alive ? n - c * L.nWinR : 0;
    const float* chn = a.pyr + int64_t(frame) * a.pyr_fs + L.off + (r * a.stride / a.shrink) + int64_t(c * a.stride / a.shrink) * L.hP;
    const float thrC = a.cascThr;
    float h = 0.f;
    if (MODE == 2)
    {
        const CascNode2* nodes = a.nodes2 + L.nodeOff;
        for (int t = 0; t < a.nTrees; t++)
        {
            if (!__any(alive))
            {
                break;
            }
            const CascNode2 nd = nodes[t]; // uniform address: scalar loads
            if (alive)
            {
                const float f0 = chn[nd.cid[0]];
                const bool lt0 = f0 < nd.thr[0];
                const float f1 = chn[lt0 ? nd.cid[1] : nd.cid[2]];
                const float th1 = lt0 ? nd.thr[1] : nd.thr[2];
                const bool lt1 = f1 < th1;
                // k after two steps: lt0 ? (lt1 ? 3 : 4) : (lt1 ? 5 : 6)
                const float hv = lt0 ? (lt1 ? nd.hs[0] : nd.hs[1]) : (lt1 ? nd.hs[2] : nd.hs[3]);
                h += hv;
                alive = h > thrC;
            }
        }
    }
    else if (MODE == 1)
    {
        const uint32_t* cid = a.cidAll + L.nodeOff;
        const int D = a.treeDepth;
        for (int t = 0; t < a.nTrees; t++)
        {
            if (!__any(alive))
            {
                break;
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
        const uint32_t* cid = a.cidAll + L.nodeOff;
        for (int t = 0; t < a.nTrees; t++)
        {
            if (!__any(alive))
            {
                break;
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
    // hit compaction: ballot + prefix count, one atomic per wave
    const unsigned long long mask = __ballot(alive);
    if (mask)
    {
        const int lane = threadIdx.x & 63;
        const int cnt = __popcll(mask);
        int base = 0;
        if (lane == (__ffsll((long long)mask) - 1))
        {
            base = atomicAdd(a.counts + frame, cnt);
        }
        base = __shfl(base, __ffsll((long long)mask) - 1);
        if (alive)
        {
            const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
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

// Sort each frame's hits into the reference's order (level, then c, then r:
// ACF.cpp:326-329, acfDetect1.cpp:86-96) by rank counting, and map them to
// image boxes (ACF.cpp:302-312).  Hit lists are small (<= maxHits), the keys
// are unique, so every hit's rank is the number of hits with a smaller key.
struct BoxLevel
{
    double shw_h, shw_w;
    int32_t bw, bh; // cvRound(modelDs / scale), precomputed on the host (ACF.cpp:304)
};

__global__ void __launch_bounds__(256) k_sort_map(const acf_hip_hit* __restrict__ hits, const int32_t* __restrict__ counts,
    int maxHits, const BoxLevel* __restrict__ bl, int stride, int shift_h, int shift_w,
    acf_hip_hit* __restrict__ sortedHits, acf_hip_detection* __restrict__ dets)
{
    const int frame = blockIdx.y;
    const int n = min(counts[frame], maxHits);
    const acf_hip_hit* H = hits + int64_t(frame) * maxHits;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const acf_hip_hit me = H[i];
        const long long key = (((long long)me.scale) << 40) | (((long long)me.c) << 20) | (long long)me.r;
        int rank = 0;
        for (int j = 0; j < n; j++)
        {
            const acf_hip_hit o = H[j];
            const long long ko = (((long long)o.scale) << 40) | (((long long)o.c) << 20) | (long long)o.r;
            rank += ko < key;
        }
        sortedHits[int64_t(frame) * maxHits + rank] = me;
        const BoxLevel b = bl[me.scale];
        acf_hip_detection d;
        // roi = ({c*stride, r*stride}); x = int(double(x + shift)/scaleshw) (truncation)
        d.x = (int)((double)(me.c * stride + shift_w) / b.shw_w);
        d.y = (int)((double)(me.r * stride + shift_h) / b.shw_h);
        d.w = b.bw;
        d.h = b.bh;
        d.score = me.score;
        d.scale = me.scale;
        dets[int64_t(frame) * maxHits + rank] = d;
    }
}

// Fixed-capacity export record per frame for the multi-GPU gather:
// [count, cap x {x,y,w,h,score bits,scale}] int32.
__global__ void __launch_bounds__(256) k_export(const acf_hip_detection* __restrict__ dets, const int32_t* __restrict__ counts,
    int maxHits, int cap, int32_t* __restrict__ dst)
{
    const int frame = blockIdx.y;
    const int n = min(min(counts[frame], maxHits), cap);
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
