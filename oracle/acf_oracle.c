/*
 * acf_oracle.c — CPU restatement of the reference's chnsPyramid + acfDetect path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under acf_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference/src/lib/acf/acf/, T/ = toolbox/).  Arithmetic is plain IEEE
 * binary32 in the reference's association order, compiled with
 * -ffp-contract=off (the reference is a plain SSE2 build: no FMA).
 *
 * Pinning status (see tests/test_oracle_vs_ref.py, oracle/Makefile):
 *  - convTri1 (incl. the in-place aliased call), convTri, grad1/gradMag's
 *    gradients, gradHist: BIT-EXACT against the reference's own toolbox
 *    sources compiled unmodified into oracle/_ref/libacfref.so.
 *  - gradMag M/O, gradMagNorm: the reference uses _mm_rsqrt_ps/_mm_rcp_ps
 *    (T/sse.hpp:185-192), ~12-bit approximations whose bits are CPU-vendor
 *    specific; this oracle uses exact 1/sqrt and 1/x at those three sites
 *    ("T-exact").  Pinned against _ref within the rcp/rsqrt bound (1.5*2^-12
 *    relative per op).
 *  - resample / rgb2luv / rgb2gray / rgb2hsv (T/imResampleMex.cpp, T/rgbConvertMex.cpp): the two files include
 *    OpenCV headers that are absent from this image; their OpenCV-free function bodies are compiled from the
 *    reference text BY LINE RANGE (oracle/Makefile) and this restatement is BIT-EXACT against them
 *    (rgb2luv_sse's U/V within its one _mm_rcp_ps: the T-exact contract above).
 *  - getScales / chnsPyramid / chnsCompute / acfDetect1 / box mapping: integer
 *    and double control logic restated from OpenCV-typed code that cannot be
 *    compiled here; the reference's tests hold no golden vectors for them
 *    (SURVEY.md §4): PARITY UNPINNED by the reference, pinned by the
 *    committed fixtures in tests/golden/ generated from this file.
 */
#include "../include/acf_hip.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ACFO_API __attribute__((visibility("default")))

static void* xmalloc(size_t n)
{
    /* 64-byte aligned like cv::Mat storage (CV_MALLOC_ALIGN): the reference's kernels choose their SSE or scalar
     * bodies by pointer alignment (T/gradientMex.cpp:262, T/rgbConvertMex.cpp:92,343), which matters in T-ref mode below. */
    void* p = aligned_alloc(64, (n ? n + 63 : 64) & ~(size_t)63);
    if (!p)
    {
        abort();
    }
    return p;
}

/* ------------------------------------------------------------------------
 * T-ref mode (build container only; tests/golden/make_tref.py, tests/test_tref_end_to_end.py).
 * The orchestration below (chnsCompute, chnsPyramid) normally calls this file's T-exact restatements of the toolbox
 * kernels.  With a table installed by acfo_set_ref_kernels it calls, at the same places and with the same arguments,
 * the reference's OWN compiled kernels (oracle/_ref/libacfref.so: rsqrtps / rcpps and all), so that a whole pyramid
 * and the detections on it can be compared between the two arithmetic tiers.  The table holds plain function
 * addresses handed in by the test (ctypes; a null entry keeps the restatement for that kernel, which is how the test
 * shows that the bit-exact stages alone reproduce the T-exact pyramid bit for bit); this file never loads _ref itself,
 * and nothing of it reaches the product.
 * ---------------------------------------------------------------------- */
typedef struct acfo_ref_kernels
{
    void (*convTri)(float* I, float* O, int h, int w, int d, int r, int s);
    void (*convTri1)(float* I, float* O, int h, int w, int d, float p, int s);
    void (*gradMag)(float* I, float* M, float* O, int h, int w, int d, int full);
    void (*gradMagNorm)(float* M, float* S, int h, int w, float norm);
    void (*gradHist)(float* M, float* O, float* H, int h, int w, int bin, int nOrients, int softBin, int full);
    void (*resample)(float* A, float* B, int ha, int hb, int wa, int wb, int d, float r);
    int (*rgbConvert)(float* I, float* J, int n, int d, int flag, float nrm);
} acfo_ref_kernels;

static __thread acfo_ref_kernels g_refk;
static __thread int g_refOn = 0;

ACFO_API void acfo_set_ref_kernels(const acfo_ref_kernels* k)
{
    g_refOn = k != NULL;
    if (k)
    {
        g_refk = *k;
    }
}

/* ------------------------------------------------------------------------
 * A third arithmetic tier, for the T-ref study only (tests/golden/make_tref.py): "T-approx".  Where the reference uses
 * _mm_rsqrt_ps / _mm_rcp_ps (three sites: gradMag, gradMagNorm, rgb2luv_sse) Intel's manual promises |relative error| <=
 * 1.5 * 2^-12 and nothing else: the bits differ between CPU vendors and generations.  acfo_set_approx(mode) makes those
 * three sites return the exact value squeezed to a 12-bit mantissa — mode 1 rounded to nearest (error <= 2^-13), mode 2
 * truncated (error < 2^-12) — two more approximations that meet that promise.  How far detections move between two such
 * conforming approximations is how far the reference's own detections move from one conforming CPU to another: the
 * yardstick for the T-ref / T-exact gap.  mode 0 (the default, everything else): the exact tier.
 * ---------------------------------------------------------------------- */
static __thread int g_approx = 0;
ACFO_API void acfo_set_approx(int mode)
{
    g_approx = mode;
}
static inline float approx12(float v)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    if ((b & 0x7f800000u) == 0x7f800000u || (b & 0x7f800000u) == 0)
    {
        return v; /* inf, NaN, zero, subnormal: as they are */
    }
    if (g_approx == 1)
    {
        b += 0x400u; /* round to nearest at bit 11 (a carry into the exponent is the right result) */
    }
    b &= 0xfffff800u; /* 12 mantissa bits kept */
    memcpy(&v, &b, 4);
    return v;
}

/* ------------------------------------------------------------------------
 * The T-ref tier as a TABLE: acfo_set_approx(3).  _mm_rcp_ps / _mm_rsqrt_ps (T/sse.hpp:185-192) are, on the CPUs probed so far
 * (tests/golden/make_x86_tables.py, which checks it for all 2^32 inputs: an Intel Xeon, family 6 model 207 — the build host — and an
 * AMD EPYC 9575F — the GPU boxes' host), pure functions of few input bits:
 *   rcp(x)   = sign | 2^(127 - e) scaled RCP[m >> 11]            4096 entries: the results for x in [1, 2), by the top 12 mantissa bits
 *   rsqrt(x) = 2^(-(e - 127 - odd) / 2) scaled RSQ[odd][m >> 11]  2 x 4096 entries: the results for x in [1, 2) and [2, 4)
 * (the Intel part decides on 11 / 10 bits: its entries repeat in pairs / fours; the AMD part uses all 12)
 * with zero / subnormal inputs -> inf of the input's sign (the instruction treats subnormals as zero), inf -> 0, NaN -> quiet NaN,
 * results below the normal range flushed to zero, rsqrt of a negative -> the default NaN (0xffc00000).  With the tables of a
 * CPU installed (acfo_set_x86_tables) the three sites below return that CPU's bits: gradMag, gradMagNorm and rgb2luv_sse become
 * BIT-EXACT against the reference's own compiled kernels on that CPU (tests/test_oracle_vs_ref.py), and the whole path reproduces
 * the reference's detections (tests/test_tref_end_to_end.py).  acfo_x86_probe reads the tables from the CPU this runs on.
 * ---------------------------------------------------------------------- */
static uint32_t g_x86Rcp[4096], g_x86Rsq[8192];
static int g_x86Set = 0;
ACFO_API void acfo_set_x86_tables(const uint32_t* rcp4096, const uint32_t* rsqrt8192)
{
    memcpy(g_x86Rcp, rcp4096, sizeof(g_x86Rcp));
    memcpy(g_x86Rsq, rsqrt8192, sizeof(g_x86Rsq));
    g_x86Set = 1;
}
ACFO_API uint32_t acfo_x86_rcp_bits(uint32_t u)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e == 0xffu)
    {
        return m ? (u | 0x400000u) : s; /* NaN quieted; 1 / inf = 0 of the same sign */
    }
    if (e == 0)
    {
        return s | 0x7f800000u; /* zero and subnormals: inf */
    }
    const uint32_t t = g_x86Rcp[m >> 11];
    const int re = (int)((t >> 23) & 0xffu) + 127 - (int)e;
    if (re <= 0)
    {
        return s; /* underflow: flushed to zero */
    }
    return s | ((uint32_t)re << 23) | (t & 0x7fffffu);
}
ACFO_API uint32_t acfo_x86_rsqrt_bits(uint32_t u)
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
    const int ue = (int)e - 127, odd = ue & 1, half = (ue - odd) / 2; /* x = 4^half * [1, 4) */
    const uint32_t t = g_x86Rsq[(odd << 12) | (m >> 11)];
    const int re = (int)((t >> 23) & 0xffu) - half;
    return ((uint32_t)re << 23) | (t & 0x7fffffu);
}
static inline float x86_rcp(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u = acfo_x86_rcp_bits(u);
    memcpy(&x, &u, 4);
    return x;
}
static inline float x86_rsqrt(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u = acfo_x86_rsqrt_bits(u);
    memcpy(&x, &u, 4);
    return x;
}
/* The approximation tiers' two operations: mode 3 the installed CPU's bits, modes 1 / 2 the exact value squeezed to 12 bits. */
static inline float ap_rcp(float x)
{
    return g_approx == 3 ? x86_rcp(x) : approx12(1.0f / x);
}
static inline float ap_rsqrt(float x)
{
    return g_approx == 3 ? x86_rsqrt(x) : approx12(1.0f / sqrtf(x));
}
/* A position-mixed 64-bit digest of both functions over the bit patterns first, first + stride, ... (count of them): how the
 * device's table functions are compared with this file's for EVERY input without moving 2^32 results (tests/test_gpu_arith.py). */
ACFO_API void acfo_x86_digest(uint32_t first, uint64_t count, uint32_t stride, uint64_t out[2])
{
    uint64_t a = 0, b = 0;
    for (uint64_t i = 0; i < count; i++)
    {
        const uint32_t u = first + (uint32_t)(i * stride);
        const uint64_t k = ((uint64_t)u * 0x9e3779b97f4a7c15ull) | 1ull;
        a += k * acfo_x86_rcp_bits(u);
        b += k * acfo_x86_rsqrt_bits(u);
    }
    out[0] = a;
    out[1] = b;
}
#if defined(__SSE__)
#include <xmmintrin.h>
static inline uint32_t hw_rcp_bits(uint32_t u)
{
    float f, o;
    memcpy(&f, &u, 4);
    o = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(f)));
    memcpy(&u, &o, 4);
    return u;
}
static inline uint32_t hw_rsqrt_bits(uint32_t u)
{
    float f, o;
    memcpy(&f, &u, 4);
    o = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(f)));
    memcpy(&u, &o, 4);
    return u;
}
/* The tables of the CPU this runs on: _mm_rcp_ps over [1, 2) and _mm_rsqrt_ps over [1, 4) at the first input of every table cell. */
ACFO_API int acfo_x86_probe(uint32_t* rcp4096, uint32_t* rsqrt8192)
{
    for (uint32_t i = 0; i < 4096; i++)
    {
        rcp4096[i] = hw_rcp_bits((127u << 23) | (i << 11));
        rsqrt8192[i] = hw_rsqrt_bits((127u << 23) | (i << 11));
        rsqrt8192[4096 + i] = hw_rsqrt_bits((128u << 23) | (i << 11));
    }
    return 1;
}
/* The installed tables against the instructions of the CPU this runs on, over first, first + stride, ...: the numbers of inputs
 * whose table result differs (0 / 0 over all 2^32 inputs is what makes the table a statement about the instruction). */
ACFO_API int acfo_x86_verify(uint32_t first, uint64_t count, uint32_t stride, uint64_t bad[2])
{
    uint64_t b0 = 0, b1 = 0;
    for (uint64_t i = 0; i < count; i++)
    {
        const uint32_t u = first + (uint32_t)(i * stride);
        b0 += hw_rcp_bits(u) != acfo_x86_rcp_bits(u);
        b1 += hw_rsqrt_bits(u) != acfo_x86_rsqrt_bits(u);
    }
    bad[0] = b0;
    bad[1] = b1;
    return 1;
}
#else
ACFO_API int acfo_x86_probe(uint32_t* rcp4096, uint32_t* rsqrt8192)
{
    (void)rcp4096;
    (void)rsqrt8192;
    return 0;
}
ACFO_API int acfo_x86_verify(uint32_t first, uint64_t count, uint32_t stride, uint64_t bad[2])
{
    (void)first;
    (void)count;
    (void)stride;
    bad[0] = bad[1] = 0;
    return 0;
}
#endif

/* ------------------------------------------------------------------------
 * a1  Detector::getScales — chnsPyramid.cpp:461-529.
 * Upright naming: H = image height (reference sz.width, because the image is
 * transposed: chnsPyramid.cpp:232, ACF.cpp:137), W = image width (sz.height).
 * minDs_h/minDs_w likewise (reference minDs.width/minDs.height, ACFIO.h:168-181).
 * ---------------------------------------------------------------------- */
ACFO_API int acfo_get_scales(int nPerOct, int nOctUp, int minDs_h, int minDs_w, int shrink, int H, int W,
    double* scales, double* shw_h, double* shw_w, int cap)
{
    if ((long)H * (long)W == 0)
    {
        return 0; /* :476-479 */
    }
    /* :481-482  util::log2(x) = log(x)/log(2) (util/acf_math.h:20-29) */
    double ratio_w = (double)H / (double)minDs_h;
    double ratio_h = (double)W / (double)minDs_w;
    double rmin = (ratio_h < ratio_w) ? ratio_h : ratio_w; /* std::min(ratio.width, ratio.height) */
    int nScales = (int)floor((double)nPerOct * ((double)nOctUp + log(rmin) / log(2.0)) + 1.0);
    if (nScales <= 0)
    {
        return 0;
    }
    /* :484-488 */
    double d0 = (double)W, d1 = (double)H;
    if (W >= H)
    {
        double t = d0;
        d0 = d1;
        d1 = t;
    }
    double* tmp = (double*)xmalloc(sizeof(double) * (size_t)(nScales + 1));
    for (int i = 0; i < nScales; i++) /* :490-510 */
    {
        double s = pow(2.0, -(double)i / (double)nPerOct + (double)nOctUp);
        double s0 = (round(d0 * s / shrink) * shrink - 0.25 * shrink) / d0;
        double s1 = (round(d0 * s / shrink) * shrink + 0.25 * shrink) / d0;
        double best_ss = 0, best_es = DBL_MAX;
        for (double j = 0.0; j < 1.0 - DBL_EPSILON; j += 0.01)
        {
            double ss = (j * (s1 - s0) + s0);
            double es0 = d0 * ss;
            es0 = fabs(es0 - round(es0 / shrink) * shrink);
            double es1 = d1 * ss;
            es1 = fabs(es1 - round(es1 / shrink) * shrink);
            double es = es0 < es1 ? es1 : es0; /* std::max(es0, es1) */
            if (es < best_es)
            {
                best_ss = ss;
                best_es = es;
            }
        }
        tmp[i] = best_ss;
    }
    tmp[nScales] = 0; /* :512-513 */
    int n = 0;
    for (int i = 1; i < nScales + 1; i++) /* :515-526 */
    {
        if (tmp[i] != tmp[i - 1])
        {
            double s = tmp[i - 1];
            if (n < cap)
            {
                scales[n] = s;
                shw_h[n] = round((double)H * s / shrink) * shrink / H; /* x: sz.width */
                shw_w[n] = round((double)W * s / shrink) * shrink / W; /* y: sz.height */
            }
            n++;
        }
    }
    free(tmp);
    return n;
}

/* ------------------------------------------------------------------------
 * a9  resampleCoef<float> — T/imResampleMex.cpp:24-121
 * ---------------------------------------------------------------------- */
typedef struct
{
    int n;
    int* yas;
    int* ybs;
    float* wts;
    int bd[2];
} coef_t;

static void resample_coef(int ha, int hb, coef_t* c, int pad)
{
    const float s = (float)hb / (float)ha, sInv = 1 / s;
    float wt, wt0 = (float)1e-3 * s;
    int ds = ha > hb;
    int nMax, n;
    c->bd[0] = c->bd[1] = 0;
    if (ds)
    {
        n = 0;
        nMax = ha + (pad > 2 ? pad : 2) * hb;
    }
    else
    {
        n = nMax = hb;
    }
    c->wts = (float*)xmalloc(sizeof(float) * (size_t)nMax);
    c->yas = (int*)xmalloc(sizeof(int) * (size_t)nMax);
    c->ybs = (int*)xmalloc(sizeof(int) * (size_t)nMax);
    if (ds)
    {
        for (int yb = 0; yb < hb; yb++) /* :47-92 */
        {
            float ya0f = yb * sInv, ya1f = ya0f + sInv, W = 0;
            int ya0 = (int)ceilf(ya0f), ya1 = (int)ya1f, n1 = 0;
            for (int ya = ya0 - 1; ya < ya1 + 1; ya++)
            {
                wt = s;
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
                    c->ybs[n] = yb;
                    c->yas[n] = ya;
                    c->wts[n] = wt;
                    n++;
                    n1++;
                    W += wt;
                }
            }
            if (W > 1)
            {
                for (int i = 0; i < n1; i++)
                {
                    c->wts[n - n1 + i] /= W;
                }
            }
            if (n1 > c->bd[0])
            {
                c->bd[0] = n1;
            }
            while (n1 < pad)
            {
                c->ybs[n] = yb;
                c->yas[n] = c->yas[n - 1];
                c->wts[n] = 0;
                n++;
                n1++;
            }
        }
    }
    else
    {
        for (int yb = 0; yb < hb; yb++) /* :96-119 */
        {
            float yaf = ((float).5 + yb) * sInv - (float).5;
            int ya = (int)floorf(yaf);
            wt = 1;
            if (ya >= 0 && ya < ha - 1)
            {
                wt = 1 - (yaf - ya);
            }
            if (ya < 0)
            {
                ya = 0;
                c->bd[0]++;
            }
            if (ya >= ha - 1)
            {
                ya = ha - 1;
                c->bd[1]++;
            }
            c->ybs[yb] = yb;
            c->yas[yb] = ya;
            c->wts[yb] = wt;
        }
    }
    c->n = n;
}

static void coef_free(coef_t* c)
{
    free(c->wts);
    free(c->yas);
    free(c->ybs);
}

/* resample<float> — T/imResampleMex.cpp:124-383.  The SSE and scalar bodies
 * of the reference compute the same expression (ADD is left-associated,
 * T/sse.hpp:135-142), so one scalar restatement covers both. */
ACFO_API int acfo_resample(const float* A, float* B, int ha, int hb, int wa, int wb, int d, float r)
{
    if (!A || !B || A == B)
    {
        return ACF_HIP_E_INVALID; /* :127-129 */
    }
    int hn, wn, x, x1 = 0, y, z, xa, xb, ya;
    const float *A0, *A1, *A2, *A3;
    float *B0, wt, wt1;
    float* C = (float*)xmalloc(sizeof(float) * (size_t)(ha + 4));
    for (y = ha; y < ha + 4; y++)
    {
        C[y] = 0;
    }
    coef_t cx, cy;
    resample_coef(wa, wb, &cx, 0);
    resample_coef(ha, hb, &cy, 4);
    wn = cx.n;
    hn = cy.n;
    (void)wn;
    int *xas = cx.yas, *xbs = cx.ybs, *yas = cy.yas, *ybs = cy.ybs;
    float *xwts = cx.wts, *ywts = cy.wts;
    int* xbd = cx.bd;
    int* ybd = cy.bd;
    if (wa == 2 * wb)
    {
        r /= 2;
    }
    if (wa == 3 * wb)
    {
        r /= 3;
    }
    if (wa == 4 * wb)
    {
        r /= 4;
    }
    r /= (float)(1 + 1e-6);
    for (y = 0; y < hn; y++)
    {
        ywts[y] *= r;
    }
    for (z = 0; z < d; z++)
    {
        for (x = 0; x < wb; x++)
        {
            if (x == 0)
            {
                x1 = 0;
            }
            xa = xas[x1];
            xb = xbs[x1];
            wt = xwts[x1];
            wt1 = 1 - wt;
            A0 = A + (size_t)z * ha * wa + (size_t)xa * ha;
            A1 = A0 + ha;
            A2 = A1 + ha;
            A3 = A2 + ha;
            B0 = B + (size_t)z * hb * wb + (size_t)xb * hb;
            /* x direction (A -> C) :190-280 */
            if (wa == 2 * wb)
            {
                for (y = 0; y < ha; y++)
                {
                    C[y] = A0[y] + A1[y];
                }
                x1 += 2;
            }
            else if (wa == 3 * wb)
            {
                for (y = 0; y < ha; y++)
                {
                    C[y] = A0[y] + A1[y] + A2[y];
                }
                x1 += 3;
            }
            else if (wa == 4 * wb)
            {
                for (y = 0; y < ha; y++)
                {
                    C[y] = A0[y] + A1[y] + A2[y] + A3[y];
                }
                x1 += 4;
            }
            else if (wa > wb)
            {
                int m = 1;
                while (x1 + m < cx.n && xb == xbs[x1 + m])
                {
                    m++;
                }
                if (m == 1)
                {
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = A0[y] * xwts[x1 + 0];
                    }
                }
                if (m == 2)
                {
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = A0[y] * xwts[x1 + 0] + A1[y] * xwts[x1 + 1];
                    }
                }
                if (m == 3)
                {
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = A0[y] * xwts[x1 + 0] + A1[y] * xwts[x1 + 1] + A2[y] * xwts[x1 + 2];
                    }
                }
                if (m >= 4)
                {
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = A0[y] * xwts[x1 + 0] + A1[y] * xwts[x1 + 1] + A2[y] * xwts[x1 + 2] + A3[y] * xwts[x1 + 3];
                    }
                }
                for (int x0 = 4; x0 < m; x0++)
                {
                    A1 = A0 + (size_t)x0 * ha;
                    wt1 = xwts[x1 + x0];
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = C[y] + A1[y] * wt1;
                    }
                }
                x1 += m;
            }
            else
            {
                int xBd = x < xbd[0] || x >= wb - xbd[1];
                x1++;
                if (xBd)
                {
                    memcpy(C, A0, sizeof(float) * (size_t)ha);
                }
                else
                {
                    for (y = 0; y < ha; y++)
                    {
                        C[y] = A0[y] * wt + A1[y] * wt1;
                    }
                }
            }
            /* y direction (C -> B) :283-373 */
            if (ha == hb * 2)
            {
                float r2 = r / 2;
                for (y = 0; y < hb; y++)
                {
                    B0[y] = (C[2 * y] + C[2 * y + 1]) * r2;
                }
            }
            else if (ha == hb * 3)
            {
                for (y = 0; y < hb; y++)
                {
                    B0[y] = (C[3 * y] + C[3 * y + 1] + C[3 * y + 2]) * (r / 3);
                }
            }
            else if (ha == hb * 4)
            {
                for (y = 0; y < hb; y++)
                {
                    B0[y] = (C[4 * y] + C[4 * y + 1] + C[4 * y + 2] + C[4 * y + 3]) * (r / 4);
                }
            }
            else if (ha > hb)
            {
                y = 0;
#define U(o) C[ya + o] * ywts[y * 4 + o]
                if (ybd[0] == 2)
                {
                    for (; y < hb; y++)
                    {
                        ya = yas[y * 4];
                        B0[y] = U(0) + U(1);
                    }
                }
                if (ybd[0] == 3)
                {
                    for (; y < hb; y++)
                    {
                        ya = yas[y * 4];
                        B0[y] = U(0) + U(1) + U(2);
                    }
                }
                if (ybd[0] == 4)
                {
                    for (; y < hb; y++)
                    {
                        ya = yas[y * 4];
                        B0[y] = U(0) + U(1) + U(2) + U(3);
                    }
                }
                if (ybd[0] > 4)
                {
                    memset(B0, 0, sizeof(float) * (size_t)hb);
                    for (; y < hn; y++)
                    {
                        B0[ybs[y]] += C[yas[y]] * ywts[y];
                    }
                }
#undef U
            }
            else
            {
                for (y = 0; y < ybd[0]; y++)
                {
                    B0[y] = C[yas[y]] * ywts[y];
                }
                for (; y < hb - ybd[1]; y++)
                {
                    B0[y] = C[yas[y]] * ywts[y] + C[yas[y] + 1] * (r - ywts[y]);
                }
                for (; y < hb; y++)
                {
                    B0[y] = C[yas[y]] * ywts[y];
                }
            }
        }
    }
    coef_free(&cx);
    coef_free(&cy);
    free(C);
    return ACF_HIP_OK;
}

/* ------------------------------------------------------------------------
 * a3  rgb2luv — T/rgbConvertMex.cpp:20-59 (setup), :62-84 (scalar),
 *               :88-190 (SSE body).  nrm = 1.
 * ---------------------------------------------------------------------- */
static float g_lTable[1064];
static int g_lInit = 0;

ACFO_API const float* acfo_luv_table(void)
{
    if (!g_lInit)
    {
        const float y0 = (float)((6.0 / 29) * (6.0 / 29) * (6.0 / 29));
        const float a = (float)((29.0 / 3) * (29.0 / 3) * (29.0 / 3));
        float maxi = (float)1.0 / 270;
        for (int i = 0; i < 1025; i++) /* :47-52 */
        {
            float y = (float)(i / 1024.0);
            float l = y > y0 ? 116 * (float)pow((double)y, 1.0 / 3.0) - 16 : y * a;
            g_lTable[i] = l * maxi;
        }
        for (int i = 1025; i < 1064; i++)
        {
            g_lTable[i] = g_lTable[i - 1];
        }
        g_lInit = 1;
    }
    return g_lTable;
}

/* I: 3 planes of n floats (R,G,B); J: 3 planes (L,U,V).  The reference takes
 * the SSE body iff n%4==0 and pointers are 16-byte aligned (:92, :343; cv::Mat
 * data is); otherwise the scalar body, which associates the denominator
 * differently and divides (:80-82).  Both are restated; rcp -> exact 1/x. */
ACFO_API void acfo_rgb2luv(const float* I, float* J, int n)
{
    const float* lTable = acfo_luv_table();
    const float z1 = 1.0f;
    float mr[3], mg[3], mb[3];
    const float un = (float)0.197833, vn = (float)0.468331;
    mr[0] = (float)0.430574 * z1;
    mr[1] = (float)0.222015 * z1;
    mr[2] = (float)0.020183 * z1;
    mg[0] = (float)0.341550 * z1;
    mg[1] = (float)0.706655 * z1;
    mg[2] = (float)0.129553 * z1;
    mb[0] = (float)0.178325 * z1;
    mb[1] = (float)0.071330 * z1;
    mb[2] = (float)0.939180 * z1;
    float maxi = (float)1.0 / 270;
    const float minu = -88 * maxi, minv = -134 * maxi;
    const float *R = I, *G = I + n, *Bp = I + 2 * (size_t)n;
    float *L = J, *U = J + n, *V = J + 2 * (size_t)n;
    if (n % 4 == 0)
    {
        const float cun = 13 * un, cvn = 13 * vn;
        for (int i = 0; i < n; i++)
        {
            float r = R[i], g = G[i], b = Bp[i];
            float x = (r * mr[0] + g * mg[0]) + b * mb[0]; /* :139 */
            float y = (r * mr[1] + g * mg[1]) + b * mb[1];
            float z = (r * mr[2] + g * mg[2]) + b * mb[2];
            float zz = 1.0f / (x + (1e-35f + (15.0f * y + 3.0f * z))); /* :161, RCP -> exact */
            if (g_approx)
            {
                zz = ap_rcp(x + (1e-35f + (15.0f * y + 3.0f * z))); /* T-approx / the table tier (acfo_set_approx) */
            }
            float lf = 1024.0f * y;                                   /* :162 */
            float u = (52.0f * x) * zz - cun;                         /* :163 */
            float v = (117.0f * y) * zz - cvn;                        /* :164 */
            float l = lTable[(int)lf];                                /* :171 */
            L[i] = l;
            U[i] = l * u - minu; /* :182 */
            V[i] = l * v - minv; /* :184 */
        }
    }
    else
    {
        for (int i = 0; i < n; i++) /* :69-83 */
        {
            float r = R[i], g = G[i], b = Bp[i];
            float x = mr[0] * r + mg[0] * g + mb[0] * b;
            float y = mr[1] * r + mg[1] * g + mb[1] * b;
            float z = mr[2] * r + mg[2] * g + mb[2] * b;
            float l = lTable[(int)(y * 1024)];
            L[i] = l;
            z = 1 / (x + 15 * y + 3 * z + (float)1e-35);
            U[i] = l * (13 * 4 * x * z - 13 * un) - minu;
            V[i] = l * (13 * 9 * y * z - 13 * vn) - minv;
        }
    }
}

/* rgb2gray — T/rgbConvertMex.cpp:241-252 (nrm = 1) */
/* rgb2hsv<float,float>, nrm = 1 — T/rgbConvertMex.cpp:194-238 */
ACFO_API void acfo_rgb2hsv(const float* I, float* J, int n)
{
    float *H = J, *S = H + n, *V = S + n;
    const float *R = I, *G = R + n, *B = G + n;
    const float nrm = 1.0f;
    for (int i = 0; i < n; i++)
    {
        const float r = R[i], g = G[i], b = B[i];
        float h, s, v, minv, maxv;
        if (r == g && g == b)
        {
            H[i] = 0;
            S[i] = 0;
            V[i] = r * nrm;
            continue;
        }
        else if (r >= g && r >= b)
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
        v = maxv * nrm;
        H[i] = h;
        S[i] = s;
        V[i] = v;
    }
}

ACFO_API void acfo_rgb2gray(const float* I, float* J, int n)
{
    const float mr = (float).2989360213 * 1.0f, mg = (float).5870430745 * 1.0f, mb = (float).1140209043 * 1.0f;
    const float *R = I, *G = I + n, *B = I + 2 * (size_t)n;
    for (int i = 0; i < n; i++)
    {
        J[i] = R[i] * mr + G[i] * mg + B[i] * mb;
    }
}

/* ------------------------------------------------------------------------
 * Image entry for 8-bit input — L/ACF.cpp:114-119 (cvt8UC3To32FC3 =
 * convertTo(CV_32FC3, 1/255)), :137,149 (I.t()), L/MatP.cpp:51-73 (plane split).
 * `in` is an upright image of H rows x W pixels, cpp bytes per pixel, rowStride
 * bytes between rows; r/g/b sit at byte offsets ro/go/bo of a pixel (the apps
 * swizzle BGR(A) to RGB before the call, A/acf/acf.cpp).  Output: nOut planes
 * float[W][H].  OpenCV's 8u->32f cvtScale works in float: float(v) * (float)(1/255.0)
 * (third-party arithmetic, SURVEY.md 8c: parity unpinned for this one product).
 * ---------------------------------------------------------------------- */
ACFO_API void acfo_ingest_u8(const uint8_t* in, int H, int W, int cpp, int ro, int go, int bo, int rowStride, float* out, int nOut)
{
    const float sc = (float)(1.0 / 255.0);
    const int off[3] = { ro, go, bo };
    for (int c = 0; c < nOut; c++)
    {
        float* P = out + (size_t)c * H * W;
        for (int x = 0; x < W; x++)
        {
            for (int y = 0; y < H; y++)
            {
                P[(size_t)x * H + y] = (float)in[(size_t)y * rowStride + (size_t)x * cpp + off[c]] * sc;
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * a4  convTri1 / convTri1Y (s == 1) — T/convConst.cpp:445-525.
 * O may alias I: that is how the pyramid calls it (chnsCompute.cpp:239,
 * chnsPyramid.cpp:404; MatP::create is a no-op for an equal shape,
 * MatP.cpp:51-73), and then column i-1 has already been overwritten with
 * output when column i is filtered (H2 in SURVEY.md).  Restated literally so
 * the aliasing falls out of the pointer arithmetic.
 * ---------------------------------------------------------------------- */
static void conv_tri1_y(const float* I, float* O, int h, float p)
{
    /* :468-489 (s == 1 branch; SSE C4 == the scalar expression) */
    int j = 0;
    O[j] = (1 + p) * I[j] + I[j + 1];
    j++;
    for (; j < h - 1; j++)
    {
        O[j] = I[j - 1] + p * I[j] + I[j + 1];
    }
    O[j] = I[j - 1] + (1 + p) * I[j];
}

ACFO_API int acfo_conv_tri1(float* I, float* O, int h, int w, int d, float p, int s)
{
    if (s != 1 || h < 2)
    {
        return ACF_HIP_E_UNSUPPORTED;
    }
    const float nrm = 1.0f / ((p + 2) * (p + 2));
    float* T = (float*)xmalloc(sizeof(float) * (size_t)h);
    for (int d0 = 0; d0 < d; d0++)
    {
        for (int i = s / 2; i < w; i += s)
        {
            float *Il, *Im, *Ir;
            Il = Im = Ir = I + (size_t)i * h + (size_t)d0 * h * w;
            if (i > 0)
            {
                Il -= h;
            }
            if (i < w - 1)
            {
                Ir += h;
            }
            for (int j = 0; j < h; j++)
            {
                T[j] = nrm * (Il[j] + p * Im[j] + Ir[j]);
            }
            conv_tri1_y(T, O, h, p);
            O += h / s;
        }
    }
    free(T);
    return ACF_HIP_OK;
}

/* ------------------------------------------------------------------------
 * a6  convTri / convTriY (s == 1) — T/convConst.cpp:269-344, 347-442
 * ---------------------------------------------------------------------- */
static void conv_tri_y(const float* I, float* O, int h, int r)
{
    r++;
    float t, u;
    int j, r0 = r - 1, r1 = r + 1, r2 = 2 * h - r, h0 = r + 1, h1 = h - r + 1, h2 = h;
    u = t = I[0];
    for (j = 1; j < r; j++)
    {
        u += t += I[j];
    }
    u = 2 * u - t;
    t = 0;
    O[0] = u;
    j = 1;
    for (; j < h0; j++)
    {
        O[j] = u += t += I[r - j] + I[r0 + j] - 2 * I[j - 1];
    }
    for (; j < h1; j++)
    {
        O[j] = u += t += I[j - r1] + I[r0 + j] - 2 * I[j - 1];
    }
    for (; j < h2; j++)
    {
        O[j] = u += t += I[j - r1] + I[r2 - j] - 2 * I[j - 1];
    }
}

ACFO_API int acfo_conv_tri(const float* I, float* O, int h, int w, int d, int r, int s)
{
    if (s != 1)
    {
        return ACF_HIP_E_UNSUPPORTED;
    }
    r++;
    float nrm = 1.0f / (r * r * r * r);
    int i, j;
    float *T = (float*)xmalloc(sizeof(float) * 2 * (size_t)h), *U = T + h;
    while (d-- > 0)
    {
        for (j = 0; j < h; j++) /* :366-400 */
        {
            U[j] = T[j] = I[j];
        }
        for (i = 1; i < r; i++)
        {
            for (j = 0; j < h; j++)
            {
                U[j] += T[j] += I[j + (size_t)i * h];
            }
        }
        for (j = 0; j < h; j++)
        {
            U[j] = nrm * (2 * U[j] - T[j]);
            T[j] = 0;
        }
        conv_tri_y(U, O, h, r - 1); /* :402-408 with s == 1 */
        O += h;
        for (i = 1; i < w; i++) /* :409-438 */
        {
            const float* Il = I + (ptrdiff_t)(i - 1 - r) * h;
            if (i <= r)
            {
                Il = I + (ptrdiff_t)(r - i) * h;
            }
            const float* Im = I + (ptrdiff_t)(i - 1) * h;
            const float* Ir = I + (ptrdiff_t)(i - 1 + r) * h;
            if (i > w - r)
            {
                Ir = I + (ptrdiff_t)(2 * w - r - i) * h;
            }
            for (j = 0; j < h; j++)
            {
                U[j] += nrm * (T[j] += Il[j] + Ir[j] - 2 * Im[j]);
            }
            conv_tri_y(U, O, h, r - 1);
            O += h;
        }
        I += (size_t)w * h;
    }
    free(T);
    return ACF_HIP_OK;
}

/* Detector::convTri dispatch — convTri.cpp:204-253 (+ convConst :144-200).
 * I may equal J (in place).  The sepFilter2D fallback (:224-251, planes
 * smaller than 4 or 2r+1 >= min dim) is unreachable from the pyramid because
 * minDs >= 4*shrink (chnsPyramid.cpp:200) and is reported as unsupported. */
ACFO_API int acfo_conv_tri_dispatch(float* I, float* J, int h, int w, int d, double r, int s)
{
    if (r == 0 && s == 1)
    {
        if (I != J)
        {
            memcpy(J, I, sizeof(float) * (size_t)h * w * d);
        }
        return ACF_HIP_OK;
    }
    int m = h < w ? h : w;
    int nomex = ((m < 4) || (2 * r + 1) >= m);
    if (nomex)
    {
        return ACF_HIP_E_UNSUPPORTED;
    }
    if ((r > 0) && (r <= 1.0) && (s <= 2))
    {
        float p = (float)(12.0 / r / (r + 2.0) - 2.0);
        if (g_refOn && g_refk.convTri1)
        {
            g_refk.convTri1(I, J, h, w, d, p, s);
            return ACF_HIP_OK;
        }
        return acfo_conv_tri1(I, J, h, w, d, p, s);
    }
    float pf = (float)r;
    int ri = (int)roundf(pf); /* convConst: int r = std::round(p) */
    if (ri >= m / 2)
    {
        return ACF_HIP_E_INVALID; /* "mask larger than image" :166-169 */
    }
    if (I == J)
    {
        return ACF_HIP_E_INVALID; /* the pyramid never calls convTri(r>1) in place */
    }
    if (g_refOn && g_refk.convTri)
    {
        g_refk.convTri(I, J, h, w, d, ri, s);
        return ACF_HIP_OK;
    }
    return acfo_conv_tri(I, J, h, w, d, ri, s);
}

/* ------------------------------------------------------------------------
 * a5  grad1 / ACosTable / gradMag — T/gradientMex.cpp:17-87,103-165,168-251
 * ---------------------------------------------------------------------- */
#define ACFO_PI 3.14159265f
#define ACOS_N 10000
#define ACOS_B 10
static float g_acos[ACOS_N * 2 + ACOS_B * 2];
static int g_acosInit = 0;

ACFO_API const float* acfo_acos_table(void)
{
    if (!g_acosInit)
    {
        float* a1 = g_acos + ACOS_N + ACOS_B;
        int i;
        const int n = ACOS_N, b = ACOS_B;
        for (i = -n - b; i < -n; i++)
        {
            a1[i] = ACFO_PI;
        }
        for (i = -n; i < n; i++)
        {
            a1[i] = acosf(i / (float)n); /* float(std::acos(float)) */
        }
        for (i = n; i < n + b; i++)
        {
            a1[i] = 0;
        }
        for (i = -n - b; i < n / 10; i++)
        {
            if (a1[i] > ACFO_PI - 1e-6f)
            {
                a1[i] = ACFO_PI - 1e-6f;
            }
        }
        g_acosInit = 1;
    }
    return g_acos; /* index with [i + ACOS_N + ACOS_B] */
}

static void grad1(const float* I, float* Gx, float* Gy, int h, int w, int x)
{
    int y;
    const float *Ip, *In;
    float r;
    Ip = I - h;
    In = I + h;
    r = .5f;
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
    for (y = 0; y < h; y++)
    {
        Gx[y] = (In[y] - Ip[y]) * r;
    }
    /* GRADY(1); Ip--; for(y=1; y<h-1; y++) GRADY(.5f); In--; GRADY(1);  :58 */
    Gy[0] = (I[1] - I[0]) * 1;
    for (y = 1; y < h - 1; y++)
    {
        Gy[y] = (I[y + 1] - I[y - 1]) * .5f;
    }
    Gy[h - 1] = (I[h - 1] - I[h - 2]) * 1;
}

/* Exported so the oracle's gradients can be pinned bit-exactly against the
 * reference's grad2 (T/gradientMex.cpp:90-101). */
ACFO_API void acfo_grad2(const float* I, float* Gx, float* Gy, int h, int w, int d)
{
    for (int c = 0; c < d; c++)
    {
        for (int x = 0; x < w; x++)
        {
            size_t o = (size_t)c * w * h + (size_t)x * h;
            grad1(I + o, Gx + o, Gy + o, h, w, x);
        }
    }
}

ACFO_API int acfo_grad_mag(const float* I, float* M, float* O, int h, int w, int d, int full)
{
    if (h < 2 || w < 2)
    {
        return ACF_HIP_E_INVALID;
    }
    const float* acosT = acfo_acos_table() + ACOS_N + ACOS_B;
    const float acMult = (float)ACOS_N;
    const float upper = (float)(ACOS_N + ACOS_B - 1), lower = -(float)(ACOS_N + ACOS_B - 1);
    float* Gx = (float*)xmalloc(sizeof(float) * (size_t)h * d);
    float* Gy = (float*)xmalloc(sizeof(float) * (size_t)h * d);
    float* M2 = (float*)xmalloc(sizeof(float) * (size_t)h * d);
    for (int x = 0; x < w; x++)
    {
        for (int c = 0; c < d; c++) /* :191-207 */
        {
            grad1(I + (size_t)x * h + (size_t)c * w * h, Gx + (size_t)c * h, Gy + (size_t)c * h, h, w, x);
            for (int y = 0; y < h; y++)
            {
                size_t y1 = (size_t)h * c + y;
                M2[y1] = Gx[y1] * Gx[y1] + Gy[y1] * Gy[y1];
                if (c == 0)
                {
                    continue;
                }
                if (M2[y1] > M2[y])
                {
                    M2[y] = M2[y1];
                    Gx[y] = Gx[y1];
                    Gy[y] = Gy[y1];
                }
            }
        }
        /* :209-219; RCPSQRT/RCP -> exact.  Written so that gcc vectorises it (IEEE sqrtps / divps, -fno-math-errno): the
         * reference's loop is SSE too, and bench.py's cpu_baseline times this function. */
        if (g_approx)
        {
            /* T-approx: RCPSQRT and RCP as 12-bit approximations (see acfo_set_approx); the rest as below */
            for (int y = 0; y < h; y++)
            {
                float m = ap_rsqrt(M2[y]);
                m = m < 1e10f ? m : 1e10f;
                M2[y] = ap_rcp(m);
                if (O)
                {
                    float g = (Gx[y] * m) * acMult;
                    uint32_t gb, yb;
                    memcpy(&gb, &g, 4);
                    memcpy(&yb, &Gy[y], 4);
                    gb ^= yb & 0x80000000u;
                    memcpy(&g, &gb, 4);
                    g = g < upper ? g : upper;
                    g = g > lower ? g : lower;
                    Gx[y] = g;
                }
            }
        }
        else if (O)
        {
            for (int y = 0; y < h; y++)
            {
                float m = 1.0f / sqrtf(M2[y]);
                m = m < 1e10f ? m : 1e10f; /* _mm_min_ps(a,b): a<b ? a : b */
                M2[y] = 1.0f / m;
                float g = (Gx[y] * m) * acMult;
                uint32_t gb, yb;
                memcpy(&gb, &g, 4);
                memcpy(&yb, &Gy[y], 4);
                gb ^= yb & 0x80000000u; /* XOR with the sign bit of Gy */
                memcpy(&g, &gb, 4);
                g = g < upper ? g : upper; /* MIN_sse */
                g = g > lower ? g : lower; /* MAX_sse */
                Gx[y] = g;
            }
        }
        else
        {
            for (int y = 0; y < h; y++)
            {
                float m = 1.0f / sqrtf(M2[y]);
                m = m < 1e10f ? m : 1e10f;
                M2[y] = 1.0f / m;
            }
        }
        memcpy(M + (size_t)x * h, M2, sizeof(float) * (size_t)h);
        if (O)
        {
            for (int y = 0; y < h; y++)
            {
                O[(size_t)x * h + y] = acosT[(int)Gx[y]];
            }
            if (full)
            {
                for (int y = 0; y < h; y++)
                {
                    O[(size_t)x * h + y] += (Gy[y] < 0) * ACFO_PI;
                }
            }
        }
    }
    free(Gx);
    free(Gy);
    free(M2);
    return ACF_HIP_OK;
}

/* a7 gradMagNorm — T/gradientMex.cpp:254-275.  Vector body M*rcp(S+norm) with
 * rcp -> exact reciprocal; the scalar tail (n%4 trailing elements, or every
 * element when M/S are not 16-byte aligned) divides.  cv::Mat planes are
 * aligned, so the tail is the last n%4 elements. */
ACFO_API void acfo_grad_mag_norm(float* M, const float* S, int h, int w, float norm)
{
    int i = 0, n = h * w, n4 = n / 4;
    if (g_approx)
    {
        for (; i < n4 * 4; i++)
        {
            M[i] = M[i] * ap_rcp(S[i] + norm); /* T-approx / the table tier (acfo_set_approx) */
        }
    }
    for (; i < n4 * 4; i++)
    {
        M[i] = M[i] * (1.0f / (S[i] + norm));
    }
    for (; i < n; i++)
    {
        M[i] /= (S[i] + norm);
    }
}

/* ------------------------------------------------------------------------
 * a8  gradQuantize + gradHist — T/gradientMex.cpp:278-372, 375-509: softBin even, >= 0 ("interpolate w.r.t. orientation
 * only", :451-509) or < 0 ("no interpolation w.r.t. either orientation or spatial bin", :391-450).  Odd softBin (trilinear
 * spatial binning, :511 on, and the 8/7 boundary normalisation :636-662 — HOG / FHOG features) is not restated.
 * ---------------------------------------------------------------------- */
ACFO_API int acfo_grad_hist(const float* M, const float* O, float* H, int h, int w, int bin, int nOrients, int softBin, int full)
{
    if (softBin % 2 != 0)
    {
        return ACF_HIP_E_UNSUPPORTED; /* trilinear (:511 on) and the 8/7 boundary normalisation of odd softBin (:636-662) */
    }
    const int interpolate = softBin >= 0;           /* gradQuantize(..., softBin >= 0), :391 */
    const int second = !(softBin < 0 && softBin % 2 == 0); /* the branch that also adds M1 into O1 (:451) */
    const int hb = h / bin, wb = w / bin, h0 = hb * bin, w0 = wb * bin, nb = wb * hb;
    const float s = (float)bin, sInv2 = 1 / s / s;
    const float oMult = (float)nOrients / (full ? 2 * ACFO_PI : ACFO_PI);
    const int oMax = nOrients * nb;
    int* O0 = (int*)xmalloc(sizeof(int) * (size_t)h);
    int* O1 = (int*)xmalloc(sizeof(int) * (size_t)h);
    float* M0 = (float*)xmalloc(sizeof(float) * (size_t)h);
    float* M1 = (float*)xmalloc(sizeof(float) * (size_t)h);
    for (int x = 0; x < w0; x++)
    {
        const float* Oc = O + (size_t)x * h;
        const float* Mc = M + (size_t)x * h;
        for (int i = 0; i < h0 && interpolate; i++) /* gradQuantize, interpolate=true :296-313,331-353 */
        {
            float o = Oc[i] * oMult;
            int o0 = (int)o;
            float od = o - o0;
            o0 *= nb;
            if (o0 >= oMax)
            {
                o0 = 0;
            }
            O0[i] = o0;
            int o1 = o0 + nb;
            if (o1 == oMax)
            {
                o1 = 0;
            }
            O1[i] = o1;
            float m = Mc[i] * sInv2;
            M1[i] = od * m;
            M0[i] = m - M1[i];
        }
        for (int i = 0; i < h0 && !interpolate; i++) /* interpolate=false :316-327,355-370 */
        {
            float o = Oc[i] * oMult;
            int o0 = (int)(o + .5f);
            o0 *= nb;
            if (o0 >= oMax)
            {
                o0 = 0;
            }
            O0[i] = o0;
            M0[i] = Mc[i] * sInv2;
            M1[i] = 0;
            O1[i] = 0;
        }
        float* H1 = H + (size_t)(x / bin) * hb; /* :454 */
        for (int y = 0; y < h0;)
        {
            for (int y1 = 0; y1 < bin; y1++)
            {
                H1[O0[y]] += M0[y];
                if (second)
                {
                    H1[O1[y]] += M1[y];
                }
                y++;
            }
            H1++;
        }
    }
    free(O0);
    free(O1);
    free(M0);
    free(M1);
    return ACF_HIP_OK;
}

/* The toolbox kernels as the orchestration below calls them: this file's T-exact restatements, or — in T-ref mode —
 * the reference's own compiled functions with the same arguments (see acfo_set_ref_kernels). */
static int st_resample(float* A, float* B, int ha, int hb, int wa, int wb, int d, float r)
{
    if (g_refOn && g_refk.resample)
    {
        g_refk.resample(A, B, ha, hb, wa, wb, d, r);
        return ACF_HIP_OK;
    }
    return acfo_resample(A, B, ha, hb, wa, wb, d, r);
}
static int st_grad_mag(float* I, float* M, float* O, int h, int w, int d, int full)
{
    if (g_refOn && g_refk.gradMag)
    {
        g_refk.gradMag(I, M, O, h, w, d, full);
        return ACF_HIP_OK;
    }
    return acfo_grad_mag(I, M, O, h, w, d, full);
}
static void st_grad_mag_norm(float* M, float* S, int h, int w, float norm)
{
    if (g_refOn && g_refk.gradMagNorm)
    {
        g_refk.gradMagNorm(M, S, h, w, norm);
        return;
    }
    acfo_grad_mag_norm(M, S, h, w, norm);
}
static int st_grad_hist(float* M, float* O, float* H, int h, int w, int bin, int nOrients, int softBin, int full)
{
    if (g_refOn && g_refk.gradHist)
    {
        g_refk.gradHist(M, O, H, h, w, bin, nOrients, softBin, full);
        return ACF_HIP_OK;
    }
    return acfo_grad_hist(M, O, H, h, w, bin, nOrients, softBin, full);
}
/* flag as rgbConvertMex.cpp:389-393: 0 gray, 2 luv, 3 hsv; nrm = 1.0f (rgbConvert.cpp:164) */
static int st_rgb_convert(float* I, float* J, int n, int flag)
{
    if (g_refOn && g_refk.rgbConvert)
    {
        return g_refk.rgbConvert(I, J, n, 3, flag, 1.0f) ? ACF_HIP_E_INVALID : ACF_HIP_OK;
    }
    if (flag == 2)
    {
        acfo_rgb2luv(I, J, n);
    }
    else if (flag == 3)
    {
        acfo_rgb2hsv(I, J, n);
    }
    else
    {
        acfo_rgb2gray(I, J, n);
    }
    return ACF_HIP_OK;
}

/* ------------------------------------------------------------------------
 * Plan: scale list, real/approx split, per-level geometry.
 * chnsPyramid.cpp:270-292 + acfDetect1.cpp:258-259.
 * ---------------------------------------------------------------------- */
static int n_chns(const acf_hip_params* p, int d_color)
{
    int n = 0;
    if (p->colorEnabled)
    {
        n += d_color;
    }
    if (p->gradMagEnabled)
    {
        n += 1;
    }
    if (p->gradHistEnabled)
    {
        n += p->nOrients;
    }
    return n;
}

/* number of colour planes after rgbConvert (rgbConvert.cpp:101-170) */
static int color_planes(const acf_hip_params* p, int d_in)
{
    if (p->colorSpace == ACF_HIP_CS_GRAY)
    {
        return 1;
    }
    (void)d_in;
    return 3;
}

ACFO_API int acfo_plan(const acf_hip_params* p, int H, int W, int d_in, acf_hip_level* lv, int cap, int* nChnsOut)
{
    double* sc = (double*)xmalloc(sizeof(double) * 3 * 4096);
    int n = acfo_get_scales(p->nPerOct, p->nOctUp, p->minDs_h, p->minDs_w, p->shrink, H, W, sc, sc + 4096, sc + 8192, 4096);
    int nC = n_chns(p, color_planes(p, d_in));
    if (nChnsOut)
    {
        *nChnsOut = nC;
    }
    if (n > cap)
    {
        free(sc);
        return n;
    }
    const int shrink = p->shrink;
    /* :272-292 (1-based isR/isN in the reference, 0-based here) */
    int* isR = (int*)xmalloc(sizeof(int) * (size_t)(n + 1));
    int nR = 0;
    for (int i = 0; i < n; i++)
    {
        if ((i % (p->nApprox + 1)) == 0)
        {
            isR[nR++] = i + 1;
        }
    }
    int* isH = (int*)xmalloc(sizeof(int) * (size_t)(nR + 2));
    for (int i = 0; i < nR + 1; i++)
    {
        isH[i] = 0;
    }
    isH[nR] = n;
    for (int i = 0; i < (nR - 1 > 0 ? nR - 1 : 0); i++)
    {
        isH[i + 1] = (isR[i] + isR[i + 1]) / 2;
    }
    int64_t off = 0;
    for (int i = 0; i < nR; i++)
    {
        for (int j = isH[i]; j < isH[i + 1]; j++)
        {
            lv[j].realIndex = isR[i] - 1;
        }
    }
    for (int i = 0; i < n; i++)
    {
        lv[i].scale = sc[i];
        lv[i].scalehw_h = sc[4096 + i];
        lv[i].scalehw_w = sc[8192 + i];
        lv[i].isReal = (i % (p->nApprox + 1)) == 0;
        /* :300 / :390: round(sz*s/shrink) */
        lv[i].hC = (int)round((double)H * sc[i] / (double)shrink);
        lv[i].wC = (int)round((double)W * sc[i] / (double)shrink);
        lv[i].hP = lv[i].hC + 2 * (p->pad_h / shrink); /* :417-420 */
        lv[i].wP = lv[i].wC + 2 * (p->pad_w / shrink);
        /* acfDetect1.cpp:258-259 */
        lv[i].nWinR = (int)ceilf((float)(lv[i].hP * shrink - p->modelDsPad_h + 1) / p->stride);
        lv[i].nWinC = (int)ceilf((float)(lv[i].wP * shrink - p->modelDsPad_w + 1) / p->stride);
        if (lv[i].nWinR < 0)
        {
            lv[i].nWinR = 0;
        }
        if (lv[i].nWinC < 0)
        {
            lv[i].nWinC = 0;
        }
        lv[i].offset = off;
        off += (int64_t)nC * lv[i].hP * lv[i].wP;
    }
    free(isR);
    free(isH);
    free(sc);
    return n;
}

/* ------------------------------------------------------------------------
 * a10  chnsCompute + addChn — chnsCompute.cpp:146-370.
 * I: d planes [w][h] (colour space already applied, "orig"); smoothed IN
 * PLACE exactly as the reference does (:239).  out: nC planes [w/shrink][h/shrink].
 * Optional taps (any may be NULL): M before normalisation, O, S, Mnorm.
 * ---------------------------------------------------------------------- */
typedef struct acfo_taps
{
    float* image;    /* d planes, before smoothing */
    float* smoothed; /* d planes */
    float* M;
    float* O;
    float* S;
    float* Mnorm;
} acfo_taps;

/* MO: NULL, or two planes [w][h] — gradient magnitude and orientation that came WITH the image (an input of five planes,
 * chnsCompute.cpp:219-226,263-269: `M = MO[0]; O = MO[1]` instead of gradientMag, normalisation included). */
static int chns_compute_mo(float* I, int h, int w, int d, const acf_hip_params* p, float* out, const acfo_taps* taps, const float* MO);
static int chns_compute(float* I, int h, int w, int d, const acf_hip_params* p, float* out, const acfo_taps* taps)
{
    return chns_compute_mo(I, h, w, d, p, out, taps, NULL);
}
static int chns_compute_mo(float* I, int h, int w, int d, const acf_hip_params* p, float* out, const acfo_taps* taps, const float* MO)
{
    const int shrink = p->shrink;
    if (h % shrink || w % shrink)
    {
        return ACF_HIP_E_UNSUPPORTED; /* crop :203-217 never triggers from chnsPyramid */
    }
    const int hs = h / shrink, ws = w / shrink;
    const size_t np = (size_t)h * w, ns = (size_t)hs * ws;
    int rc;
    if (taps && taps->image)
    {
        memcpy(taps->image, I, sizeof(float) * np * d);
    }
    /* :235-239 rgbConvert "orig" = no-op; convTri(I, I, pColor.smooth, 1) in place */
    rc = acfo_conv_tri_dispatch(I, I, h, w, d, p->colorSmooth, 1);
    if (rc)
    {
        return rc;
    }
    if (taps && taps->smoothed)
    {
        memcpy(taps->smoothed, I, sizeof(float) * np * d);
    }
    float* o = out;
    if (p->colorEnabled) /* :253-256 addChn -> imResample(.., 1.0) :346-351 */
    {
        if (shrink == 1)
        {
            memcpy(o, I, sizeof(float) * np * d);
        }
        else
        {
            rc = st_resample(I, o, h, hs, w, ws, d, 1.0f);
            if (rc)
            {
                return rc;
            }
        }
        o += ns * d;
    }
    if (!(p->gradMagEnabled || p->gradHistEnabled))
    {
        return ACF_HIP_OK;
    }
    /* :263-308 */
    float* M = (float*)xmalloc(sizeof(float) * np);
    float* O = (float*)xmalloc(sizeof(float) * np);
    if (p->colorChn < 0 || p->colorChn >= d)
    {
        free(M);
        free(O);
        return ACF_HIP_E_INVALID;
    }
    if (MO) /* chnsCompute.cpp:265-269: the planes that came with the image; no gradientMag, no normalisation */
    {
        memcpy(M, MO, sizeof(float) * np);
        memcpy(O, MO + np, sizeof(float) * np);
    }
    else
    {
        rc = st_grad_mag(I + np * (size_t)p->colorChn, M, O, h, w, 1, p->full); /* gradientMag.cpp:90-98: d = 1 */
    }
    if (rc)
    {
        free(M);
        free(O);
        return rc;
    }
    if (taps && taps->M)
    {
        memcpy(taps->M, M, sizeof(float) * np);
    }
    if (taps && taps->O)
    {
        memcpy(taps->O, O, sizeof(float) * np);
    }
    if (p->normRad != 0 && !MO) /* gradientMag.cpp:120-132 */
    {
        float* S = (float*)xmalloc(sizeof(float) * np);
        rc = acfo_conv_tri_dispatch(M, S, h, w, 1, (double)p->normRad, 1);
        if (rc)
        {
            free(S);
            free(M);
            free(O);
            return rc;
        }
        st_grad_mag_norm(M, S, h, w, (float)p->normConst);
        if (taps && taps->S)
        {
            memcpy(taps->S, S, sizeof(float) * np);
        }
        free(S);
    }
    if (taps && taps->Mnorm)
    {
        memcpy(taps->Mnorm, M, sizeof(float) * np);
    }
    if (p->gradMagEnabled) /* :303-307 */
    {
        if (shrink == 1)
        {
            memcpy(o, M, sizeof(float) * np);
        }
        else
        {
            rc = st_resample(M, o, h, hs, w, ws, 1, 1.0f);
        }
        o += ns;
    }
    if (!rc && p->gradHistEnabled) /* :310-333 */
    {
        int binSize = p->binSize ? p->binSize : shrink;
        if (binSize != shrink)
        {
            rc = ACF_HIP_E_UNSUPPORTED;
        }
        else
        {
            memset(o, 0, sizeof(float) * ns * (size_t)p->nOrients); /* gradientHist.cpp:94-95 */
            rc = st_grad_hist(M, O, o, h, w, binSize, p->nOrients, p->softBin, p->full);
        }
    }
    free(M);
    free(O);
    return rc;
}

ACFO_API int acfo_chns_compute(float* I, int h, int w, int d, const acf_hip_params* p, float* out, const acfo_taps* taps)
{
    return chns_compute(I, h, w, d, p, out, taps);
}
/* the same with the image's own M, O planes (two planes [w][h]; chnsCompute.cpp:219-226) */
ACFO_API int acfo_chns_compute_mo(float* I, int h, int w, int d, const acf_hip_params* p, float* out, const float* MO)
{
    return chns_compute_mo(I, h, w, d, p, out, NULL, MO);
}

/* cv::copyMakeBorder(BORDER_REFLECT) on one plane [w][h] -> [w+2px][h+2py]
 * (chnsPyramid.cpp:410-424; MatP.cpp:122-129).  fedcba|abcdefgh|hgfedcb */
static inline int reflect(int i, int n)
{
    while (i < 0 || i >= n)
    {
        if (i < 0)
        {
            i = -i - 1;
        }
        else
        {
            i = 2 * n - 1 - i;
        }
    }
    return i;
}

static void pad_reflect(const float* src, float* dst, int h, int w, int py, int px)
{
    const int hP = h + 2 * py, wP = w + 2 * px;
    for (int x = 0; x < wP; x++)
    {
        int sx = reflect(x - px, w);
        for (int y = 0; y < hP; y++)
        {
            int sy = reflect(y - py, h);
            dst[(size_t)x * hP + y] = src[(size_t)sx * h + sy];
        }
    }
}

/* ------------------------------------------------------------------------
 * a2 + a11  Detector::chnsPyramid — chnsPyramid.cpp:160-456.
 * frame: d_in planes [W][H].  The caller's frame is NOT modified (the
 * reference smooths the caller's planes in place through shallow copies when
 * isLuv; here a private copy takes that role, which yields the same outputs).
 * levels: from acfo_plan.  out: one frame's fused pyramid (level offsets from
 * the plan).  taps: per real-scale ordinal, may be NULL.  chns_taps: per
 * level, unsmoothed/unpadded channels, may be NULL.
 * ---------------------------------------------------------------------- */
/* ------------------------------------------------------------------------
 * f1  bbNms (types max / maxg) + ObjectDetector::prune — bbNms.cpp:111-192 (nmsMax), :229-304 (threshold, dispatch),
 * ObjectDetector.cpp:28-44 (prune).  boxes [n][4] = {x, y, w, h}; scores f64.  keep receives the indices of the
 * survivors in output order; returns their number.  type: 1 max, 2 maxg (0: everything, in input order).
 * The reference orders by score with std::sort (util/ordered.h:23-31): the order among equal scores is unspecified
 * there; here (and on the device) ties keep their input order.  Checked against the reference's own numbers only
 * through the properties its code implies (tests/test_nms.py): PARITY UNPINNED by a reference-produced vector.
 * ---------------------------------------------------------------------- */
ACFO_API int acfo_nms(const int32_t* boxes, const double* scores, int n, int type, int ovrDnmUnion, double overlap, double thr,
    int doPrune, int maxCount, double pruneRatio, int32_t* keep)
{
    if (n <= 0)
    {
        return 0;
    }
    int* ord = (int*)xmalloc(sizeof(int) * (size_t)n);
    int m = 0;
    if (type == 0)
    {
        for (int i = 0; i < n; i++)
        {
            keep[i] = i;
        }
        free(ord);
        m = n;
    }
    else
    {
        for (int i = 0; i < n; i++) /* :276-279 erase score < thr */
        {
            if (!(scores[i] < thr))
            {
                ord[m++] = i;
            }
        }
        /* descending score, stable (insertion sort: n is small in the tests) */
        for (int i = 1; i < m; i++)
        {
            const int v = ord[i];
            int j = i - 1;
            while (j >= 0 && scores[ord[j]] < scores[v])
            {
                ord[j + 1] = ord[j];
                j--;
            }
            ord[j + 1] = v;
        }
        char* kp = (char*)xmalloc((size_t)m + 1);
        memset(kp, 1, (size_t)m + 1);
        const int greedy = type == 2;
        for (int i = 0; i < m; i++) /* :145-183 */
        {
            if (greedy && !kp[i])
            {
                continue;
            }
            const int32_t* bi = boxes + 4 * (size_t)ord[i];
            const int xsi = bi[0], ysi = bi[1], xei = bi[0] + bi[2], yei = bi[1] + bi[3], asi = bi[2] * bi[3];
            for (int j = i + 1; j < m; j++)
            {
                if (!kp[j])
                {
                    continue;
                }
                const int32_t* bj = boxes + 4 * (size_t)ord[j];
                const int xsj = bj[0], ysj = bj[1], xej = bj[0] + bj[2], yej = bj[1] + bj[3], asj = bj[2] * bj[3];
                const int iw = (xei < xej ? xei : xej) - (xsi > xsj ? xsi : xsj);
                if (iw <= 0)
                {
                    continue;
                }
                const int ih = (yei < yej ? yei : yej) - (ysi > ysj ? ysi : ysj);
                if (ih <= 0)
                {
                    continue;
                }
                double o = (double)(iw * ih);
                const double u = ovrDnmUnion ? ((double)(asi + asj) - o) : (double)(asi < asj ? asi : asj);
                o /= u;
                if (o > overlap)
                {
                    kp[j] = 0;
                }
            }
        }
        int k = 0;
        for (int i = 0; i < m; i++)
        {
            if (kp[i])
            {
                keep[k++] = ord[i];
            }
        }
        free(kp);
        free(ord);
        m = k;
    }
    if (doPrune && m > 1) /* ObjectDetector.cpp:30-42 */
    {
        int cutoff = 1;
        const int L = maxCount < m ? maxCount : m;
        for (int i = 1; i < L; i++)
        {
            cutoff = i + 1;
            if (scores[keep[i]] < scores[keep[0]] * pruneRatio)
            {
                break;
            }
        }
        m = cutoff;
    }
    return m;
}

/* ------------------------------------------------------------------------
 * a12  image-specific lambdas — chnsPyramid.cpp:341-374, MatP.cpp:97-111 (sum, numel), acf_math.h:20-29 (util::log2 =
 * log(x) / log(2)).  sum(MatP) adds the per-plane cv::sum(plane)[0] in plane order; cv::sum accumulates the f32 data in
 * f64 in OpenCV's own SIMD-blocked order, which cannot be restated without OpenCV (absent from this image): PARITY
 * UNPINNED for that order.  The order used here and by the device (k_plane_sums) is 256 interleaved partial sums
 * combined by a binary tree; any order of the same f64 additions agrees to a few ulp of a double (1e-16 relative).
 * ---------------------------------------------------------------------- */
ACFO_API double acfo_plane_sum(const float* x, int n)
{
    double part[256];
    for (int t = 0; t < 256; t++)
    {
        double acc = 0.0;
        for (int i = t; i < n; i += 256)
        {
            acc += (double)x[i];
        }
        part[t] = acc;
    }
    for (int s = 128; s >= 1; s >>= 1)
    {
        for (int t = 0; t < s; t++)
        {
            part[t] += part[t + s];
        }
    }
    return part[0];
}

/* The two real levels the lambdas are estimated from (0-based), :343-355; returns 0 if there are fewer than two
 * candidates (CV_Assert(is.size() >= 2) in the reference). */
ACFO_API int acfo_lambda_levels(const acf_hip_params* p, int nScales, int* i0, int* i1)
{
    int is[3], n = 0;
    for (int i = 1 + p->nOctUp * p->nPerOct; i <= nScales && n < 3; i += p->nApprox + 1)
    {
        is[n++] = i - 1;
    }
    if (n < 2)
    {
        return 0;
    }
    *i0 = n > 2 ? is[1] : is[0];
    *i1 = n > 2 ? is[2] : is[1];
    return 1;
}

/* lambda of one channel type from its plane sums at the two levels — :356-373 */
ACFO_API double acfo_lambda(double sum0, double numel0, double sum1, double numel1, double scale0, double scale1)
{
    const double f0 = sum0 / numel0, f1 = sum1 / numel1;
    return -(log(f0 / f1) / log(2.0)) / (log(scale0 / scale1) / log(2.0));
}

static __thread double g_last_lambdas[3]; /* lambdas used by the last acfo_chns_pyramid call (supplied or estimated) */
ACFO_API void acfo_last_lambdas(double* out)
{
    memcpy(out, g_last_lambdas, sizeof(g_last_lambdas));
}

ACFO_API int acfo_chns_pyramid(const float* frame, int H, int W, int d_in, const acf_hip_params* p,
    const acf_hip_level* lv, int nScales, float* out, const acfo_taps* taps, float* const* chns_taps)
{
    int lam0 = -1, lam1 = -1;
    if (p->nLambdas != 3 && p->nApprox > 0 && !acfo_lambda_levels(p, nScales, &lam0, &lam1))
    {
        return ACF_HIP_E_INVALID; /* CV_Assert(is.size() >= 2), :351 */
    }
    if (p->softBin % 2 != 0)
    {
        return ACF_HIP_E_UNSUPPORTED; /* trilinear spatial binning (odd softBin): not restated */
    }
    const int shrink = p->shrink;
    const size_t np0 = (size_t)H * W;
    int rc = ACF_HIP_OK;
    /* :230-263 colour conversion once at full resolution */
    int d = 0;
    float* I = NULL; /* current source image for real scales ("I" in the reference) */
    int Ih = H, Iw = W;
    {
        const int cs = p->colorSpace;
        float* pI;
        int dI = d_in;
        if (d_in == 1 && (cs == ACF_HIP_CS_GRAY || cs == ACF_HIP_CS_ORIG)) /* :234-244 replicate to 3 planes */
        {
            pI = (float*)xmalloc(sizeof(float) * np0 * 3);
            for (int k = 0; k < 3; k++)
            {
                memcpy(pI + np0 * k, frame, sizeof(float) * np0);
            }
            dI = 3;
        }
        else if (d_in == 3 || d_in == 5) /* five planes: M, O come with the image (:248-255): they stay aside until the first real scale */
        {
            pI = (float*)xmalloc(sizeof(float) * np0 * 3);
            memcpy(pI, frame, sizeof(float) * np0 * 3);
            dI = 3;
        }
        else
        {
            return ACF_HIP_E_INVALID;
        }
        /* rgbConvert(pI, I, cs, true, isLuv) — rgbConvert.cpp:101-170 */
        if (cs == ACF_HIP_CS_ORIG || cs == ACF_HIP_CS_RGB || (p->isLuv && cs == ACF_HIP_CS_LUV))
        {
            I = pI;
            d = dI;
        }
        else if (cs == ACF_HIP_CS_LUV)
        {
            I = (float*)xmalloc(sizeof(float) * np0 * 3);
            rc = st_rgb_convert(pI, I, (int)np0, 2);
            free(pI);
            d = 3;
        }
        else if (cs == ACF_HIP_CS_HSV)
        {
            if (p->isLuv || d_in == 1)
            {
                free(pI);
                return ACF_HIP_E_INVALID; /* CV_Assert(flag == 2) :150-155; CV_Assert(flag == 0) for one plane :140-148 */
            }
            I = (float*)xmalloc(sizeof(float) * np0 * 3);
            rc = st_rgb_convert(pI, I, (int)np0, 3);
            free(pI);
            d = 3;
        }
        else if (cs == ACF_HIP_CS_GRAY)
        {
            if (p->isLuv)
            {
                free(pI);
                return ACF_HIP_E_INVALID; /* CV_Assert(flag == 2) :150-155 */
            }
            I = (float*)xmalloc(sizeof(float) * np0);
            rc = st_rgb_convert(pI, I, (int)np0, 0);
            free(pI);
            d = 1;
        }
        else
        {
            free(pI);
            return ACF_HIP_E_UNSUPPORTED;
        }
    }
    int nC = n_chns(p, d);
    /* unsmoothed, unpadded channels per level */
    float** data = (float**)xmalloc(sizeof(float*) * (size_t)nScales);
    for (int i = 0; i < nScales; i++)
    {
        data[i] = (float*)xmalloc(sizeof(float) * (size_t)nC * lv[i].hC * lv[i].wC);
    }
    /* :294-338 real scales, sequential.  Buffer ownership mimics the
     * reference's shallow copies: I1 == I when sz == sz1 (so chnsCompute's
     * in-place smoothing mutates I), and I = I1 when s == 0.5. */
    int realOrd = 0;
    for (int i = 0; i < nScales && !rc; i++)
    {
        if (!lv[i].isReal)
        {
            continue;
        }
        double s = lv[i].scale;
        int h1 = (int)round((double)H * s / (double)shrink) * shrink;
        int w1 = (int)round((double)W * s / (double)shrink) * shrink;
        float* I1;
        int I1_is_I = 0;
        /* NB: the reference compares against sz = Iin.size() (the ORIGINAL
         * size, :232,:303), not the size of the current I. */
        if (H == h1 && W == w1)
        {
            if (Ih != H || Iw != W)
            {
                /* I was replaced by a half-size image yet sz == sz1: the
                 * reference would hand a wrong-sized I to chnsCompute; this
                 * cannot happen because scales are strictly decreasing. */
                rc = ACF_HIP_E_INVALID;
                break;
            }
            I1 = I;
            I1_is_I = 1;
        }
        else
        {
            I1 = (float*)xmalloc(sizeof(float) * (size_t)h1 * w1 * d);
            rc = st_resample(I, I1, Ih, h1, Iw, w1, d, 1.0f);
            if (rc)
            {
                free(I1);
                break;
            }
        }
        if ((s == 0.5) && ((p->nApprox > 0) || (p->nPerOct == 1))) /* :313-316 */
        {
            if (!I1_is_I)
            {
                free(I);
                I = I1;
                Ih = h1;
                Iw = w1;
                I1_is_I = 1;
            }
        }
        if (d_in == 5 && realOrd == 0)
        {
            /* :318-322: the image's M, O planes ride with the FIRST real scale only — which must be the image's own size (the reference
             * pushes the full-size planes onto I1 whatever its size) */
            rc = (h1 == H && w1 == W) ? chns_compute_mo(I1, h1, w1, d, p, data[i], taps ? &taps[realOrd] : NULL, frame + 3 * np0) : ACF_HIP_E_UNSUPPORTED;
        }
        else
        {
            rc = chns_compute(I1, h1, w1, d, p, data[i], taps ? &taps[realOrd] : NULL);
        }
        if (!I1_is_I)
        {
            free(I1);
        }
        realOrd++;
    }
    /* :341-374 lambdas: the model's, or estimated from this image */
    double lambdas[3] = { 0, 0, 0 };
    if (p->nLambdas == 3)
    {
        memcpy(lambdas, p->lambdas, sizeof(lambdas));
    }
    else if (p->nApprox > 0 && !rc)
    {
        const int nTypeCh[3] = { p->colorEnabled ? d : 0, p->gradMagEnabled ? 1 : 0, p->gradHistEnabled ? p->nOrients : 0 };
        size_t z0 = 0;
        for (int j = 0; j < 3; j++)
        {
            if (!nTypeCh[j])
            {
                continue;
            }
            double s0 = 0, s1 = 0;
            const size_t c0 = (size_t)lv[lam0].hC * lv[lam0].wC, c1 = (size_t)lv[lam1].hC * lv[lam1].wC;
            for (int k = 0; k < nTypeCh[j]; k++) /* sum(MatP): per-plane sums added in plane order */
            {
                s0 += acfo_plane_sum(data[lam0] + (z0 + k) * c0, (int)c0);
                s1 += acfo_plane_sum(data[lam1] + (z0 + k) * c1, (int)c1);
            }
            lambdas[j] = acfo_lambda(s0, (double)nTypeCh[j] * c0, s1, (double)nTypeCh[j] * c1, lv[lam0].scale, lv[lam1].scale);
            z0 += nTypeCh[j];
        }
    }
    memcpy(g_last_lambdas, lambdas, sizeof(lambdas));
    /* :385-397 approximated scales */
    for (int i = 0; i < nScales && !rc; i++)
    {
        if (lv[i].isReal)
        {
            continue;
        }
        const int iR = lv[i].realIndex;
        const int hb = lv[i].hC, wb = lv[i].wC, ha = lv[iR].hC, wa = lv[iR].wC;
        const int nTypeCh[3] = { p->colorEnabled ? d : 0, p->gradMagEnabled ? 1 : 0, p->gradHistEnabled ? p->nOrients : 0 };
        size_t offA = 0, offB = 0;
        for (int j = 0; j < 3 && !rc; j++)
        {
            if (!nTypeCh[j])
            {
                continue;
            }
            double ratio = pow(lv[i].scale / lv[iR].scale, -lambdas[j]);
            rc = st_resample(data[iR] + offA, data[i] + offB, ha, hb, wa, wb, nTypeCh[j], (float)ratio);
            offA += (size_t)nTypeCh[j] * ha * wa;
            offB += (size_t)nTypeCh[j] * hb * wb;
        }
    }
    /* :399-435 smooth (in place, per type = per plane), pad, concat */
    for (int i = 0; i < nScales && !rc; i++)
    {
        const int hC = lv[i].hC, wC = lv[i].wC;
        if (chns_taps && chns_taps[i])
        {
            memcpy(chns_taps[i], data[i], sizeof(float) * (size_t)nC * hC * wC);
        }
        rc = acfo_conv_tri_dispatch(data[i], data[i], hC, wC, nC, p->smooth, 1);
        if (rc)
        {
            break;
        }
        float* dst = out + lv[i].offset;
        const int py = p->pad_h / shrink, px = p->pad_w / shrink;
        for (int c = 0; c < nC; c++)
        {
            pad_reflect(data[i] + (size_t)c * hC * wC, dst + (size_t)c * lv[i].hP * lv[i].wP, hC, wC, py, px);
        }
    }
    for (int i = 0; i < nScales; i++)
    {
        free(data[i]);
    }
    free(data);
    free(I);
    return rc;
}

/* Classifier::thrsU8 — L/ACFIOArchive.h:96-99: thrs.convertTo(thrsU8, CV_8UC1, 255.0f).  OpenCV's 32f->8u
 * cvtScale: saturate_cast<uchar>(src * (float)alpha + 0) with cvRound (round half to even).  Third-party
 * arithmetic: parity unpinned for this one conversion. */
ACFO_API void acfo_thrs_u8(const float* thrs, int n, uint8_t* out)
{
    for (int i = 0; i < n; i++)
    {
        float v = thrs[i] * 255.0f;
        /* cvRound = cvtss2si: round half to even; NaN and |v| >= 2^31 give INT_MIN, which saturates to 0 */
        long r = (v > -2147483648.0f && v < 2147483648.0f) ? lrintf(v) : -2147483647L - 1;
        out[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

/* ------------------------------------------------------------------------
 * a13-a15  createDetector / ParallelDetectionBody / acfDetect1 —
 * T/acfDetect1.cpp:72-166, 231-335, 390-406 (column-major, no rois).
 * chns: nC planes [wP][hP].  u8 != 0 selects the uint8_t body (chns/thrs are
 * then uint8_t arrays, :157-166,187-192).
 * ---------------------------------------------------------------------- */
static float evaluate_window(const void* chns1v, int u8, const uint32_t* cids, const acf_hip_params* p, const void* thrsv, float cascThr)
{
    const uint32_t* fids = p->fids;
    const float* hs = p->hs;
    const uint32_t* child = p->child;
    const int nTrees = p->nTrees, nTreeNodes = p->nTreeNodes, kDepth = p->treeDepth;
    const float* cf = (const float*)chns1v;
    const uint8_t* cu = (const uint8_t*)chns1v;
    const float* tf = (const float*)thrsv;
    const uint8_t* tu = (const uint8_t*)thrsv;
    float h = 0.f;
    uint32_t isZero = (kDepth == 0);
    for (int t = 0; t < nTrees; t++) /* :123-138 */
    {
        uint32_t offset = (uint32_t)t * (uint32_t)nTreeNodes, k = offset, k0 = (k * isZero);
        if (kDepth > 0)
        {
            for (int i = 0; i < kDepth; i++) /* getChild :100-107 */
            {
                uint32_t index = cids[fids[k]];
                float ftr = u8 ? (float)cu[index] : cf[index];
                float thr = u8 ? (float)tu[k] : tf[k];
                k = (ftr < thr) ? 1 : 2;
                k0 = k += k0 * 2;
                k += offset;
            }
        }
        else
        {
            while (child[k]) /* :146-166 */
            {
                uint32_t index = cids[fids[k]];
                float ftr = u8 ? (float)cu[index] : cf[index];
                float thr = u8 ? (float)tu[k] : tf[k];
                k = (ftr < thr) ? 1 : 0;
                k0 = k = child[k0] - k + offset;
            }
        }
        h += hs[k];
        if (h <= cascThr)
        {
            break;
        }
    }
    return h;
}

ACFO_API int acfo_acf_detect1(const void* chns, int u8, const void* thrs, int hP, int wP, int nChns, const acf_hip_params* p,
    acf_hip_hit* out, int cap, int scale_tag)
{
    const int shrink = p->shrink, stride = p->stride;
    const int modelHt = p->modelDsPad_h, modelWd = p->modelDsPad_w; /* after the swap :252-256 */
    const int height = hP, width = wP;
    const int height1 = (int)ceilf((float)(height * shrink - modelHt + 1) / stride);
    const int width1 = (int)ceilf((float)(width * shrink - modelWd + 1) / stride);
    const int mW = modelWd / shrink, mH = modelHt / shrink;
    const int rowStride = hP;
    uint32_t* cids = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)nChns * mW * mH);
    {
        int m = 0, area = width * height; /* :390-406 */
        for (int z = 0; z < nChns; z++)
        {
            for (int c = 0; c < mW; c++)
            {
                for (int r = 0; r < mH; r++)
                {
                    cids[m++] = (uint32_t)(z * area + c * height + r);
                }
            }
        }
    }
    const float cascThr = (float)p->cascThr; /* detector->cascThr = cascThr (float member) :323 */
    int n = 0;
    for (int c = 0; c < width1; c++) /* :84-98 */
    {
        for (int r = 0; r < height1; r++)
        {
            int offset = (r * stride / shrink) + (c * stride / shrink) * rowStride;
            const void* base = u8 ? (const void*)((const uint8_t*)chns + offset) : (const void*)((const float*)chns + offset);
            float h = evaluate_window(base, u8, cids, p, thrs, cascThr);
            if (h > cascThr)
            {
                if (n < cap)
                {
                    out[n].scale = scale_tag;
                    out[n].c = c;
                    out[n].r = r;
                    out[n].score = h;
                }
                n++;
            }
        }
    }
    free(cids);
    return n;
}

/* Detector::evaluate(const MatP&, shrink, modelDsPad, stride) — T/acfDetect1.cpp:337-342: createDetector, cascThr = 0,
 * evaluate(0, 0): the score reached by the window at the origin (trees added until h <= cascThr). */
ACFO_API float acfo_evaluate(const float* chns, int hP, int wP, int nChns, const acf_hip_params* p, float cascThr)
{
    const int mW = p->modelDsPad_w / p->shrink, mH = p->modelDsPad_h / p->shrink;
    uint32_t* cids = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)nChns * mW * mH);
    int m = 0, area = wP * hP;
    for (int z = 0; z < nChns; z++)
    {
        for (int c = 0; c < mW; c++)
        {
            for (int r = 0; r < mH; r++)
            {
                cids[m++] = (uint32_t)(z * area + c * hP + r);
            }
        }
    }
    const float h = evaluate_window(chns, 0, cids, p, p->thrs, cascThr);
    free(cids);
    return h;
}

/* ------------------------------------------------------------------------
 * a18  LDCF post-stage (BASELINE cfg 5).  NO REFERENCE COUNTERPART (README.rst:8): restates the upstream toolbox's
 * acfDetectImg — per level C(:,:,j) = conv2(chns(:,:,mod(j-1,nC)+1), fs(:,:,j), 'same'); P.data{i} = imResample(C, .5);
 * cascade with shrink*2.  Planes are [w][h] with h contiguous (= MATLAB's column-major (h, w)); filter tap (dy, dx) of
 * filter f, channel c at filt[((f*nC + c)*5 + dx)*5 + dy].  Tap order (this repo's own choice): dx ascending, dy
 * ascending, ONE chain of fused multiply-adds from 0 in f32 (acc = fmaf(v, w, acc): each step rounds once).  Parity unpinned by construction.
 * ---------------------------------------------------------------------- */
static int round_half_away(double v)
{
    return (int)(v < 0 ? -floor(-v + 0.5) : floor(v + 0.5));
}

ACFO_API void acfo_ldcf_conv(const float* in, float* out, int h, int w, const float* f)
{
    for (int x = 0; x < w; x++)
    {
        for (int y = 0; y < h; y++)
        {
            float acc = 0.f;
            for (int dx = -2; dx <= 2; dx++)
            {
                for (int dy = -2; dy <= 2; dy++)
                {
                    const int xx = x - dx, yy = y - dy;
                    const float v = (xx >= 0 && xx < w && yy >= 0 && yy < h) ? in[(size_t)xx * h + yy] : 0.f;
                    acc = fmaf(v, f[(dx + 2) * 5 + (dy + 2)], acc); /* one chain of fused multiply-adds (exactly rounded: == the device's v_fma_f32) */
                }
            }
            out[(size_t)x * h + y] = acc;
        }
    }
}

/* Level table of the LDCF pyramid: hP' = round(.5*hP) (imResample.m), window grid with shrink*2. */
ACFO_API int64_t acfo_ldcf_plan(const acf_hip_params* p, const acf_hip_level* lv, int nScales, int nChns, acf_hip_level* out)
{
    int64_t off = 0;
    const int shrink2 = 2 * p->shrink;
    for (int i = 0; i < nScales; i++)
    {
        out[i] = lv[i];
        out[i].hP = round_half_away(0.5 * lv[i].hP);
        out[i].wP = round_half_away(0.5 * lv[i].wP);
        out[i].hC = out[i].hP;
        out[i].wC = out[i].wP;
        int n1 = (int)ceilf((float)(out[i].hP * shrink2 - p->modelDsPad_h + 1) / p->stride);
        int n2 = (int)ceilf((float)(out[i].wP * shrink2 - p->modelDsPad_w + 1) / p->stride);
        out[i].nWinR = n1 > 0 ? n1 : 0;
        out[i].nWinC = n2 > 0 ? n2 : 0;
        out[i].offset = off;
        off += (int64_t)nChns * p->ldcfK * out[i].hP * out[i].wP;
    }
    return off;
}

ACFO_API int acfo_ldcf_pyramid(const float* pyr, const acf_hip_params* p, const acf_hip_level* lv, const acf_hip_level* lvL, int nScales,
    int nChns, float* out)
{
    for (int i = 0; i < nScales; i++)
    {
        const int h = lv[i].hP, w = lv[i].wP, hb = lvL[i].hP, wb = lvL[i].wP;
        float* C = (float*)xmalloc(sizeof(float) * (size_t)h * w);
        for (int f = 0; f < p->ldcfK; f++)
        {
            for (int c = 0; c < nChns; c++)
            {
                acfo_ldcf_conv(pyr + lv[i].offset + (size_t)c * h * w, C, h, w, p->ldcfFilters + ((size_t)f * nChns + c) * 25);
                int rc = acfo_resample(C, out + lvL[i].offset + ((size_t)f * nChns + c) * hb * wb, h, hb, w, wb, 1, 1.0f);
                if (rc)
                {
                    free(C);
                    return rc;
                }
            }
        }
        free(C);
    }
    return 0;
}

/* Mean number of trees evaluated per window (diagnostic for the synthetic
 * model calibration; no reference counterpart). */
ACFO_API double acfo_mean_trees(const float* chns, int hP, int wP, int nChns, const acf_hip_params* p)
{
    const int shrink = p->shrink, stride = p->stride;
    const int height1 = (int)ceilf((float)(hP * shrink - p->modelDsPad_h + 1) / stride);
    const int width1 = (int)ceilf((float)(wP * shrink - p->modelDsPad_w + 1) / stride);
    const int mW = p->modelDsPad_w / shrink, mH = p->modelDsPad_h / shrink;
    if (height1 <= 0 || width1 <= 0 || p->treeDepth <= 0)
    {
        return 0;
    }
    const float cascThr = (float)p->cascThr;
    double total = 0;
    for (int c = 0; c < width1; c++)
    {
        for (int r = 0; r < height1; r++)
        {
            const float* c1 = chns + (r * stride / shrink) + (c * stride / shrink) * hP;
            float h = 0;
            int t;
            for (t = 0; t < p->nTrees; t++)
            {
                uint32_t offset = (uint32_t)t * p->nTreeNodes, k0 = 0, k = offset;
                for (int i = 0; i < p->treeDepth; i++)
                {
                    uint32_t f = p->fids[k];
                    uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH;
                    float ftr = c1[(size_t)z * hP * wP + cc * hP + rr];
                    k = (ftr < p->thrs[k]) ? 1 : 2;
                    k0 = k += k0 * 2;
                    k += offset;
                }
                h += p->hs[k];
                if (h <= cascThr)
                {
                    t++;
                    break;
                }
            }
            total += t;
        }
    }
    return total / ((double)height1 * width1);
}

/* ------------------------------------------------------------------------
 * a16  Detector::operator()(const Pyramid&) without NMS — ACF.cpp:268-367.
 * pyr: one frame's fused pyramid.  Returns the number of detections.
 * ---------------------------------------------------------------------- */
ACFO_API int acfo_detect(const float* pyr, const acf_hip_params* p, const acf_hip_level* lv, int nScales, int nChns,
    acf_hip_detection* out, acf_hip_hit* hits_out, int cap)
{
    /* shift = (modelDsPad - modelDs)/2 - pad  (:275; cv::Size integer ops) */
    const int shift_h = (p->modelDsPad_h - p->modelDs_h) / 2 - p->pad_h;
    const int shift_w = (p->modelDsPad_w - p->modelDs_w) / 2 - p->pad_w;
    acf_hip_hit* hits = (acf_hip_hit*)xmalloc(sizeof(acf_hip_hit) * (size_t)(cap > 0 ? cap : 1));
    int n = 0;
    for (int i = 0; i < nScales; i++)
    {
        int room = cap - n > 0 ? cap - n : 0;
        int k = acfo_acf_detect1(pyr + lv[i].offset, 0, p->thrs, lv[i].hP, lv[i].wP, nChns, p, hits, room, i);
        for (int j = 0; j < k && j < room; j++)
        {
            /* acfDetect1.cpp:326-333: roi = ({c*stride, r*stride}, winSize) then swap;
             * ACF.cpp:302-312: scale up, then swap back. */
            int rx = hits[j].c * p->stride, ry = hits[j].r * p->stride;
            /* cv::Size size(cv::Size2d(modelDs) / scale): saturate_cast<int> = cvRound = lrint */
            int sh = (int)lrint((double)p->modelDs_h / lv[i].scale);
            int sw = (int)lrint((double)p->modelDs_w / lv[i].scale);
            acf_hip_detection dd;
            dd.y = (int)((double)(ry + shift_h) / lv[i].scalehw_h);
            dd.x = (int)((double)(rx + shift_w) / lv[i].scalehw_w);
            dd.h = sh;
            dd.w = sw;
            dd.score = hits[j].score;
            dd.scale = i;
            if (out)
            {
                out[n + j] = dd;
            }
            if (hits_out)
            {
                hits_out[n + j] = hits[j];
            }
        }
        n += k;
    }
    free(hits);
    return n;
}

/* ------------------------------------------------------------------------
 * f4: the apps' resize-to-minimum-object-width (src/app/acf/acf.cpp:117-148 `Resizer`, GPUDetectionPipeline.cpp:250-266):
 *   scale = float(winSize.width) / float(minWidth);  cv::resize(image, reduced, {}, scale, scale, scale < 1 ? INTER_AREA : INTER_LINEAR)
 * on the packed 8-bit RGB image, the detector on `reduced`, boxes back with cv::Rect2f(o) * (1.f / scale) -> cv::Rect.
 * cv::resize is OpenCV (a hunter dependency, absent from /root/reference and from this image): PARITY UNPINNED.  What is
 * restated here is the published algorithm of OpenCV's imgproc/resize.cpp for CV_8U (3.4 / 4.x, the generic C++ paths, whose SIMD
 * forms are written to give the same bytes):
 *  - dsize = (cvRound(cols * fx), cvRound(rows * fy)); scale_x = 1 / fx (the caller's fx, not recomputed from the sizes)
 *  - INTER_AREA, both scales integral: exactly 2 x 2 -> (a + b + c + d + 2) >> 2; otherwise cvRound(int_sum * float(1 / area))
 *  - INTER_AREA, fractional: float tables of overlap fractions (computeResizeAreaTab), per source row a horizontal
 *    accumulation buf += S * alpha (k ascending), vertically sum = beta * buf for the first row, sum += beta * buf after, cvRound
 *  - INTER_LINEAR (and INTER_AREA when a scale < 1 would enlarge): 11-bit fixed-point weights saturate_cast<short>(w * 2048),
 *    horizontal S0 * a0 + S1 * a1 (int), vertical (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
 * cvRound = round half to even (lrint in the default rounding mode).
 * ------------------------------------------------------------------------ */
static int cv_round(double v)
{
    return (int)lrint(v);
}
static int cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}
static int cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}
static uint8_t sat_u8_f(float v)
{
    int i = (int)lrintf(v);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}
static short sat_s16_f(float v)
{
    int i = (int)lrintf(v);
    return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

ACFO_API void acfo_resize_dims(int rows, int cols, double fx, double fy, int* drows, int* dcols)
{
    *dcols = cv_round(cols * fx);
    *drows = cv_round(rows * fy);
}

typedef struct
{
    int si, di;
    float alpha;
} AreaTap;

/* computeResizeAreaTab (cn = 1: indices in pixels) */
static int area_tab(int ssize, int dsize, double scale, AreaTap* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++)
    {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3)
        {
            tab[k].di = dx;
            tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++)
        {
            tab[k].di = dx;
            tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3)
        {
            double a = fsx2 - sx2;
            a = a < 1. ? a : 1.;
            a = a < cellWidth ? a : cellWidth;
            tab[k].di = dx;
            tab[k].si = sx2;
            tab[k++].alpha = (float)(a / cellWidth);
        }
    }
    return k;
}

/* interp: 1 = INTER_LINEAR, 3 = INTER_AREA (OpenCV's enum values).  src [rows][stride bytes] with cn interleaved channels,
 * dst tight [drows][dcols][cn]; drows / dcols from acfo_resize_dims. */
ACFO_API int acfo_resize_u8(const uint8_t* src, int rows, int cols, int cn, int stride, double fx, double fy, int interp, uint8_t* dst, int drows, int dcols)
{
    if (!src || !dst || rows < 1 || cols < 1 || cn < 1 || cn > 4 || drows < 1 || dcols < 1 || fx <= 0 || fy <= 0)
    {
        return ACF_HIP_E_INVALID;
    }
    if (stride <= 0)
    {
        stride = cols * cn;
    }
    const double scale_x = 1. / fx, scale_y = 1. / fy;
    const int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
    const int area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (interp == 1 && area_fast && iscale_x == 2 && iscale_y == 2)
    {
        interp = 3; /* "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to INTER_LINEAR" */
    }
    if (interp == 3 && scale_x >= 1 && scale_y >= 1)
    {
        if (area_fast)
        {
            const int area = iscale_x * iscale_y;
            const float sc = 1.f / (float)area;
            for (int dy = 0; dy < drows; dy++)
            {
                for (int dx = 0; dx < dcols; dx++)
                {
                    for (int c = 0; c < cn; c++)
                    {
                        uint8_t* D = dst + ((size_t)dy * dcols + dx) * cn + c;
                        const int sy0 = dy * iscale_y, sx0 = dx * iscale_x;
                        if (sy0 + iscale_y > rows || sx0 + iscale_x > cols)
                        {
                            /* ResizeAreaFast_Invoker: columns >= w = ssize.width / scale_x and rows past the source take the partial-cell form */
                            int sum = 0, count = 0;
                            for (int sy = 0; sy < iscale_y; sy++)
                            {
                                if (sy0 + sy >= rows)
                                {
                                    break;
                                }
                                for (int sx = 0; sx < iscale_x; sx++)
                                {
                                    if (sx0 + sx >= cols)
                                    {
                                        break;
                                    }
                                    sum += src[(size_t)(sy0 + sy) * stride + (size_t)(sx0 + sx) * cn + c];
                                    count++;
                                }
                            }
                            *D = count ? sat_u8_f((float)sum / count) : 0;
                            continue;
                        }
                        int sum = 0;
                        for (int sy = 0; sy < iscale_y; sy++)
                        {
                            for (int sx = 0; sx < iscale_x; sx++)
                            {
                                sum += src[(size_t)(sy0 + sy) * stride + (size_t)(sx0 + sx) * cn + c];
                            }
                        }
                        *D = (iscale_x == 2 && iscale_y == 2) ? (uint8_t)((sum + 2) >> 2) : sat_u8_f((float)sum * sc);
                    }
                }
            }
            return ACF_HIP_OK;
        }
        AreaTap* xtab = (AreaTap*)xmalloc(sizeof(AreaTap) * ((size_t)cols * 2 + 2));
        AreaTap* ytab = (AreaTap*)xmalloc(sizeof(AreaTap) * ((size_t)rows * 2 + 2));
        const int nx = area_tab(cols, dcols, scale_x, xtab), nyt = area_tab(rows, drows, scale_y, ytab);
        float* buf = (float*)xmalloc(sizeof(float) * (size_t)dcols * cn);
        float* sum = (float*)xmalloc(sizeof(float) * (size_t)dcols * cn);
        int prev_dy = ytab[0].di;
        for (int i = 0; i < dcols * cn; i++)
        {
            sum[i] = 0.f;
        }
        for (int j = 0; j < nyt; j++)
        {
            const float beta = ytab[j].alpha;
            const int dy = ytab[j].di;
            const uint8_t* S = src + (size_t)ytab[j].si * stride;
            for (int i = 0; i < dcols * cn; i++)
            {
                buf[i] = 0.f;
            }
            for (int k = 0; k < nx; k++)
            {
                const float alpha = xtab[k].alpha;
                for (int c = 0; c < cn; c++)
                {
                    buf[xtab[k].di * cn + c] += S[xtab[k].si * cn + c] * alpha;
                }
            }
            if (dy != prev_dy)
            {
                uint8_t* D = dst + (size_t)prev_dy * dcols * cn;
                for (int i = 0; i < dcols * cn; i++)
                {
                    D[i] = sat_u8_f(sum[i]);
                    sum[i] = beta * buf[i];
                }
                prev_dy = dy;
            }
            else
            {
                for (int i = 0; i < dcols * cn; i++)
                {
                    sum[i] += beta * buf[i];
                }
            }
        }
        {
            uint8_t* D = dst + (size_t)prev_dy * dcols * cn;
            for (int i = 0; i < dcols * cn; i++)
            {
                D[i] = sat_u8_f(sum[i]);
            }
        }
        free(xtab);
        free(ytab);
        free(buf);
        free(sum);
        return ACF_HIP_OK;
    }
    /* INTER_LINEAR (also INTER_AREA that would enlarge) */
    int* xofs = (int*)xmalloc(sizeof(int) * (size_t)dcols);
    short* ialpha = (short*)xmalloc(sizeof(short) * 2 * (size_t)dcols);
    int xmax = dcols;
    for (int dx = 0; dx < dcols; dx++)
    {
        float f = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(f);
        f -= sx;
        if (sx < 0)
        {
            f = 0, sx = 0;
        }
        if (sx + 1 >= cols)
        {
            xmax = xmax < dx ? xmax : dx;
            if (sx >= cols - 1)
            {
                f = 0, sx = cols - 1;
            }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_s16_f((1.f - f) * 2048.f);
        ialpha[2 * dx + 1] = sat_s16_f(f * 2048.f);
    }
    for (int dy = 0; dy < drows; dy++)
    {
        float f = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(f);
        f -= sy;
        const short b0 = sat_s16_f((1.f - f) * 2048.f), b1 = sat_s16_f(f * 2048.f);
        int r0 = sy < 0 ? 0 : (sy > rows - 1 ? rows - 1 : sy);
        int r1 = sy + 1 < 0 ? 0 : (sy + 1 > rows - 1 ? rows - 1 : sy + 1);
        const uint8_t *S0 = src + (size_t)r0 * stride, *S1 = src + (size_t)r1 * stride;
        for (int dx = 0; dx < dcols; dx++)
        {
            const int sx = xofs[dx];
            for (int c = 0; c < cn; c++)
            {
                int h0, h1;
                if (dx < xmax)
                {
                    h0 = S0[sx * cn + c] * ialpha[2 * dx] + S0[(sx + 1) * cn + c] * ialpha[2 * dx + 1];
                    h1 = S1[sx * cn + c] * ialpha[2 * dx] + S1[(sx + 1) * cn + c] * ialpha[2 * dx + 1];
                }
                else
                {
                    h0 = S0[sx * cn + c] * 2048;
                    h1 = S1[sx * cn + c] * 2048;
                }
                dst[((size_t)dy * dcols + dx) * cn + c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs);
    free(ialpha);
    return ACF_HIP_OK;
}

/* Resizer::operator()(objects): o = cv::Rect2f(o) * (1.f / scale) -> cv::Rect (acf.cpp:134-143, 548-551): float products,
 * Rect_<int>(Rect_<float>) = saturate_cast<int> of each field = cvRound. */
ACFO_API void acfo_unscale_rect(float scale, const int* in, int* out)
{
    const float inv = 1.f / scale;
    for (int k = 0; k < 4; k++)
    {
        out[k] = cv_round((double)((float)in[k] * inv));
    }
}
