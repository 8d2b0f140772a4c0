// kernels_output.hip.h — part of kernels.hip.h (included from there, in its order, and nowhere else: the parts share kernels.hip.h's
// includes, its layout / arithmetic contract and the helpers of the parts before them).
// bbNms + prune on the device, the export of the gather records.
#pragma once

namespace acfhip
{

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
