// kernels_levels.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// imResample (table driven), the up-sampling form, threshold-rank cells, and the fused level kernels (approximated-scale resample + final smoothing + placement in the padded pyramid).
#pragma once

namespace acfhip
{

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

} // namespace acfhip
