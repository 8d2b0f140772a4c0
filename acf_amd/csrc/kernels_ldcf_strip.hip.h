// kernels_ldcf_strip.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// LDCF filters (cfg 5), imResample on LDS tiles (k_ldcf_tile's halving) and the strip march of the down-sampling image resamples (k_resample_strip).
#pragma once

namespace acfhip
{

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
        if (HAVEB && stepB) // (uniform.  Past B's last step / row tile its records are clamped to B's last column, whose taps lie
        {                   // outside the tile in LDS: nothing of B is stored in such a step, so nothing is computed either)
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

} // namespace acfhip
