// kernels_cascade.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// acfDetect1: the staged global-memory cascade, the LDS-tiled cascades (depth 2: k_cascade_tile3; other depths: k_cascade_tileD / tile3D), the tail (leaf codes + k_tail_scan), hit expansion, sort + box mapping.
#pragma once

namespace acfhip
{

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
    TILE_STAMP(6);
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

} // namespace acfhip
