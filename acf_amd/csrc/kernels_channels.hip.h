// kernels_channels.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// chnsCompute on a real scale: the vector image smoothing with its fused consumers (colour channels, half-size image, gradMag, convTri's x pass), the lambdas' plane sums, the border fill, gradMag, convTri (x and y passes), gradMagNorm + gradHist + addChn.
#pragma once

namespace acfhip
{

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
    const uint32_t* x86; // ARITH (option "arith"): the CPU tables gradMag's rsqrt / rcp come from
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
#define X86_TABLE_N 12288 // rcp 4096 + rsqrt 2 x 4096 (kernels.hip.h)
// ... followed in the device buffer by gradMag's own form of them (acf_hip_set_x86_tables builds it): X86_GM_N pairs {RSQ[i], rcp(RSQ[i])}
// and the bits of rcp(1e10) — see gm_inv_x86g
#define X86_GM_N 8192
#define X86_BUF_N (X86_TABLE_N + 2 * X86_GM_N + 8)
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
// The reference's own text at this site (gradientMex.cpp:209-210: m = MIN(RCPSQRT(M2), 1e10); M = RCP(m)) with one x86 CPU's
// instructions from its tables (kernels.hip.h: x86_rsqrt / x86_rcp): option "arith".
__device__ __forceinline__ void gm_inv_x86(float m2, float& m, float& M, const uint32_t* __restrict__ T)
{
    const float t = x86_rsqrt(m2, T);
    m = t < 1e10f ? t : 1e10f; // _mm_min_ps(a, b): a < b ? a : b
    M = x86_rcp(m, T);
}
// The same two results for the column kernels, whose time is their instruction stream: ONE 8-byte table read instead of two dependent
// 4-byte ones.  m2 = gx * gx + gy * gy is +0, positive or NaN — never negative — and for a normal m2 = 4^half * [1, 4) with table index i
//   t = rsqrt(m2) = 2^-half * RSQ[i]        and, when t < 1e10,        rcp(t) = 2^half * rcp(RSQ[i]):
// rcp decides on the top 12 mantissa bits of its input, which are all RSQ[i] has, and its exponent arithmetic is exact — so G[i] =
// {RSQ[i], rcp(RSQ[i])} (built on the host from the same tables) gives both with an exponent adjustment.  Everything else takes
// constants: zero / subnormal (t = inf) and NaN (the min returns its second operand) -> m = 1e10, M = rcp(1e10) = K; t >= 1e10 the
// same; m2 = inf -> t = 0, M = rcp(0) = inf.  The same floats as gm_inv_x86 for every m2 that is not negative (checked on the device for
// all 2^31 of them against gm_inv_x86: acf_hip_selftest_x86's third digest, tests/test_gpu_arith.py).
__device__ __forceinline__ void gm_inv_x86g(float m2, float& m, float& M, const uint2* __restrict__ G, float K)
{
    const uint32_t u = __float_as_uint(m2);
    const bool normal = (u - 0x00800000u) < 0x7f000000u; // [0x00800000, 0x7f7fffff]: positive normal (a set sign bit = NaN here)
    const uint32_t e1 = u >> 23;
    const uint32_t i = (((e1 & 1u) ^ 1u) << 12) | ((u >> 11) & 0xfffu); // odd = (e - 127) & 1
    const uint32_t hs = (((e1 + 1u) >> 1) - 64u) << 23;                  // half = (e - 127 - odd) / 2, in the exponent's place
    const uint2 q = G[i];
    const float t = __uint_as_float(q.x - hs);
    const bool in = normal && t < 1e10f;
    m = in ? t : 1e10f;
    M = in ? __uint_as_float(q.y + hs) : K;
    const bool isInf = u == 0x7f800000u;
    m = isInf ? 0.f : m;
    M = isInf ? __uint_as_float(0x7f800000u) : M;
}
// gm_inv_x86g against gm_inv_x86 over the bit patterns first + i * stride (as m2 >= 0 or NaN): number of patterns where m or M differ
__global__ void __launch_bounds__(256) k_gm_x86_selftest(const uint32_t* __restrict__ T, uint32_t first, unsigned long long count, uint32_t stride,
    unsigned long long* __restrict__ bad)
{
    const uint2* G = reinterpret_cast<const uint2*>(T + X86_TABLE_N);
    const float K = __uint_as_float(T[X86_TABLE_N + 2 * X86_GM_N]);
    unsigned long long nb = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256)
    {
        const uint32_t bits = first + uint32_t(i * stride);
        const float x = __uint_as_float(bits);
        if ((bits & 0x80000000u) && !((bits & 0x7f800000u) == 0x7f800000u && (bits & 0x7fffffu)))
        {
            continue; // negative and not NaN: outside gradMag's domain
        }
        float m0, M0, m1, M1;
        gm_inv_x86(x, m0, M0, T);
        gm_inv_x86g(x, m1, M1, G, K);
        nb += __float_as_uint(m0) != __float_as_uint(m1) || __float_as_uint(M0) != __float_as_uint(M1);
    }
    if (nb)
    {
        atomicAdd(&bad[0], nb);
    }
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
template <bool FULL, bool HALF, bool SHRINK, bool GRAD = false, bool TRIX = false, bool ARITH = false>
__device__ __forceinline__ void smooth_vec_body(const SmoothVecArgs& a, float* lds, int z, const float* acosT = nullptr, const uint2* x86G = nullptr, float x86K = 0.f)
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
    float gLo = -10009.0f, gHi = 10009.0f; // (SV_GRAD's clamp: VOP3 takes no literal on gfx950)
    ACF_PIN_V(gLo);
    ACF_PIN_V(gHi);
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
            if (ARITH)                                                                            \
            {                                                                                     \
                gm_inv_x86g(m2, m, mo[k], x86G, x86K);                                            \
            }                                                                                     \
            else                                                                                  \
            {                                                                                     \
                gm_inv_fast(m2, m, mo[k]);                                                        \
            }                                                                                     \
            float g = (gx * m) * 10000.0f;                                                        \
            g = __int_as_float(__float_as_int(g) ^ (__float_as_int(gy) & 0x80000000));            \
            /* g < 10009 ? g : 10009, then > -10009 (gradientMex.cpp:224-226) as ONE v_med3_f32: the two selects compile to  */ \
            /* v_min / v_max with a canonicalising v_max in front of each (5 instructions per pixel of a chain that is bound */ \
            /* by its instruction stream); g is finite (|gx| * m * 1e4 with m <= 1e10), where the median is the clamp         */ \
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(gLo), "v"(gHi));                   \
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
            const float il = prev[k]; /* Il = Im at i == 0 (convConst.cpp:503-507; a later segment's warm-up starts the same way): `prev` is the chain's first input column when it starts */ \
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
    prev[0] = c0[0].x, prev[1] = c0[0].y, prev[2] = c0[0].z, prev[3] = c0[0].w; // (the first step's left tap is its middle tap)
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
template <bool HALF, bool ARITH = false>
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
    uint2* x86G = reinterpret_cast<uint2*>(acosL + GM_ACOS_N); // ARITH: gradMag's table pairs behind the acos table (64 KB more LDS: the host asks for it)
    float x86K = 0.f;
    if (ARITH)
    {
        const uint2* gsrc = reinterpret_cast<const uint2*>(a.x86 + X86_TABLE_N);
        for (int i = threadIdx.x; i < X86_GM_N; i += blockDim.x)
        {
            x86G[i] = gsrc[i];
        }
        x86K = __uint_as_float(a.x86[X86_TABLE_N + 2 * X86_GM_N]);
    }
    __syncthreads();
    if ((fullMask >> z) & 1u)
    {
        smooth_vec_body<true, HALF, true, true, false, ARITH>(a, lds, z, acosL + 10010, x86G, x86K);
    }
    else
    {
        smooth_vec_body<false, HALF, true, true, false, ARITH>(a, lds, z, acosL + 10010, x86G, x86K);
    }
}

// k_smooth_grad with convTri's x pass over M on the same chain (smooth_vec_body's TRIX): M is written once and not read back by a
// separate x pass (8.3 MB per 1080p frame and one launch less per scale).  One segment per plane; up to 8 waves (1920 rows): the
// ring of sixteen M columns costs 64 registers, which a workgroup of ten waves cannot have.
template <bool HALF, bool ARITH = false>
__global__ void __launch_bounds__(512) k_smooth_grad_tri(SmoothVecArgs a, uint32_t fullMask)
{
    extern __shared__ float lds[]; // (k_smooth_grad's)
    const int z = a.plane0;
    float* acosL = lds + 2 * SV_MAXW * 2 * SV_K * 4;
    for (int i = threadIdx.x; i < GM_ACOS_N; i += blockDim.x)
    {
        acosL[i] = a.acos[i];
    }
    uint2* x86G = reinterpret_cast<uint2*>(acosL + GM_ACOS_N); // ARITH: gradMag's table pairs behind the acos table (64 KB more LDS: the host asks for it)
    float x86K = 0.f;
    if (ARITH)
    {
        const uint2* gsrc = reinterpret_cast<const uint2*>(a.x86 + X86_TABLE_N);
        for (int i = threadIdx.x; i < X86_GM_N; i += blockDim.x)
        {
            x86G[i] = gsrc[i];
        }
        x86K = __uint_as_float(a.x86[X86_TABLE_N + 2 * X86_GM_N]);
    }
    __syncthreads();
    if ((fullMask >> z) & 1u)
    {
        smooth_vec_body<true, HALF, true, true, true, ARITH>(a, lds, z, acosL + 10010, x86G, x86K);
    }
    else
    {
        smooth_vec_body<false, HALF, true, true, true, ARITH>(a, lds, z, acosL + 10010, x86G, x86K);
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
    const float* __restrict__ acosBase, int h, int w, int full, int64_t in_fs, int64_t out_fs, int stripsPerBlock, const uint32_t* __restrict__ x86)
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
            float m = x86 ? x86_rsqrt(m2, x86) : 1.0f / sqrtf(m2);
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
                M[o] = x86 ? x86_rcp(m, x86) : 1.0f / m;
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
template <bool BL, bool ARITH = false>
__global__ void __launch_bounds__(GMV_BLOCK) k_grad_mag_vec(const float* __restrict__ in, float* __restrict__ M, float* __restrict__ O,
    const float* __restrict__ acosBase, int h, int w, int full, int64_t in_fs, int64_t out_fs, int nFrames, int nyb, const uint32_t* __restrict__ x86 = nullptr)
{
    __shared__ float acosL[GM_ACOS_N];
    __shared__ uint2 x86G[ARITH ? X86_GM_N : 1]; // (option "arith": gradMag's table pairs, 64 KB beside the 80 KB acos table: still one workgroup per CU)
    float x86K = 0.f;
    for (int i = threadIdx.x; i < GM_ACOS_N; i += GMV_BLOCK)
    {
        acosL[i] = acosBase[i];
    }
    if (ARITH)
    {
        const uint2* gsrc = reinterpret_cast<const uint2*>(x86 + X86_TABLE_N);
        for (int i = threadIdx.x; i < X86_GM_N; i += GMV_BLOCK)
        {
            x86G[i] = gsrc[i];
        }
        x86K = __uint_as_float(x86[X86_TABLE_N + 2 * X86_GM_N]);
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
                if (ARITH)
                {
                    gm_inv_x86g(m2, m, mo[k], x86G, x86K); // option "arith": the reference's rsqrtps / rcpps bits
                }
                else
                {
                    gm_inv_fast(m2, m, mo[k]); // m = min(1 / sqrt(m2), 1e10), M = 1 / m: the IEEE results, see gm_inv_fast
                }
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
    const uint32_t* x86; // option "arith": gradMagNorm's reciprocal from the CPU tables (k_chns: when not null; k_triy_chns<.., true>: its ARITH form)
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
template <int MAXO, bool UT, bool ARITH = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_triy_chns(const float* __restrict__ Ui, ChnsArgs ca, int64_t ufs, int nyb)
{
    __shared__ float ty_lds[4][UT ? 1 : 2][64 * TY_PITCH];
    // STG: the cells of FOUR steps (16 cell rows of the wave's 16 cell columns, per channel) are collected in LDS and leave as 64
    // contiguous bytes per cell column and channel — a step alone writes 16-byte pieces, which L2 evicts before the next steps
    // complete their lines: 14.7 MB written per 1080p frame for 4.8 MB of cells (WRITE_SIZE, profiles/r05_ab/triy_nt_level_fm.txt)
#ifdef ACF_TRIY_DIRECT
    constexpr bool STG = false;
#else
    constexpr bool STG = MAXO <= 6;
#endif
    constexpr int CB_P = 20, NCB = MAXO + 1; // row pitch of a (channel, cell column) of the buffer; slots: the magnitude, the bins
    __shared__ float ty_cb[4][STG ? NCB * 16 * CB_P : 1];
    __shared__ uint32_t ty_rcp[ARITH ? 4096 : 1]; // option "arith": the rcp table of the CPU (16 KB: two workgroups per CU still fit)
    if (ARITH)
    {
        for (int i = threadIdx.x; i < 4096; i += 256)
        {
            ty_rcp[i] = ca.x86[i];
        }
        __syncthreads();
    }
    int cs = 0, cr0 = 0; // steps in the buffer; cell row of its row 0
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
                /* gradMagNorm: M * rcp(S + norm); ARITH: the rcp of one x86 CPU (n % 4 == 0 here: no scalar tail) */ \
                mn[xx][yy] = ARITH ? mr_[yy] * x86_rcp(sq[xx][yy] + ca.normConst, ty_rcp) : mr_[yy] * (1.0f / (sq[xx][yy] + ca.normConst)); \
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
                const float cv_ = (((C_[0] + C_[1]) + C_[2]) + C_[3]) * ca.rq_y;                            \
                if (STG)                                                                                    \
                {                                                                                           \
                    ty_cb[wv][xcL * CB_P + 4 * cs + ycL] = cv_;                                             \
                }                                                                                           \
                else                                                                                        \
                {                                                                                           \
                    outc[int64_t(chMag) * cellsN] = cv_;                                                    \
                }                                                                                           \
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
                    if (STG)                                                                                \
                    {                                                                                       \
                        ty_cb[wv][((1 + b) * 16 + xcL) * CB_P + 4 * cs + ycL] = H_[b];                      \
                    }                                                                                       \
                    else if (b < nO)                                                                        \
                    {                                                                                       \
                        outc[int64_t(chH_ + b) * cellsN] = H_[b];                                           \
                    }                                                                                       \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        if (STG)                                                                                            \
        {                                                                                                   \
            cs++;                                                                                           \
            if (cs == 4)                                                                                    \
            {                                                                                               \
                TY_FLUSH();                                                                                 \
            }                                                                                               \
        }                                                                                                   \
    }
    // the buffer's rows leave: lane (cell column xcL, k = lane & 3) takes rows 4k .. 4k + 3 of every channel as one 16-byte store
    // (8-byte aligned when the cell column's start is: hc need not be a multiple of 4); rows past the plane's last stay behind
#define TY_FLUSH()                                                                                          \
    {                                                                                                       \
        __builtin_amdgcn_wave_barrier();                                                                    \
        const int rows_ = min(4 * cs, hc - cr0), k4_ = 4 * ycL;                                             \
        if (x0 + 4 * xcL < w && k4_ < rows_)                                                                \
        {                                                                                                   \
            float* dst_ = chn + (uint32_t((x0 >> 2) + xcL) * uint32_t(hc) + uint32_t(cr0 + k4_));           \
            const int nv_ = min(4, rows_ - k4_);                                                            \
            const int chH_ = chMag + (ca.magEnabled ? 1 : 0);                                               \
            _Pragma("unroll") for (int sl = 0; sl < NCB; sl++)                                              \
            {                                                                                               \
                if (sl == 0 ? ca.magEnabled != 0 : (ca.histEnabled && sl - 1 < nO))                         \
                {                                                                                           \
                    const float4 v_ = *reinterpret_cast<const float4*>(&ty_cb[wv][(sl * 16 + xcL) * CB_P + k4_]); \
                    float* d_ = dst_ + int64_t(sl == 0 ? chMag : chH_ + sl - 1) * cellsN;                   \
                    if (nv_ == 4)                                                                           \
                    {                                                                                       \
                        typedef float f4u_ __attribute__((ext_vector_type(4), aligned(4)));                 \
                        f4u_ o4_ = { v_.x, v_.y, v_.z, v_.w };                                              \
                        *reinterpret_cast<f4u_*>(d_) = o4_;                                                 \
                    }                                                                                       \
                    else                                                                                    \
                    {                                                                                       \
                        d_[0] = v_.x;                                                                       \
                        if (nv_ > 1)                                                                        \
                        {                                                                                   \
                            d_[1] = v_.y;                                                                   \
                        }                                                                                   \
                        if (nv_ > 2)                                                                        \
                        {                                                                                   \
                            d_[2] = v_.z;                                                                   \
                        }                                                                                   \
                    }                                                                                       \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                    \
        cr0 += 4 * cs;                                                                                      \
        cs = 0;                                                                                             \
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
    if (STG && cs > 0)
    {
        TY_FLUSH();
    }
#undef TY_FLUSH
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
                    m = a.x86 ? m * x86_rcp(s + a.normConst, a.x86) : m * (1.0f / (s + a.normConst));
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

} // namespace acfhip
