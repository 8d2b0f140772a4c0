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
// rounded 1/sqrt and 1/x here ("T-exact"), or — option "arith" = 1 — the bits
// of one x86 CPU's instructions from its tables (x86_rcp / x86_rsqrt below).
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
// The reference's arithmetic at its three approximate sites (gradientMex.cpp:209-219,266; rgbConvertMex.cpp:161; sse.hpp:185-192).
// _mm_rcp_ps / _mm_rsqrt_ps are functions of (sign, exponent / its parity, the top 12 mantissa bits) on the CPUs probed — an Intel Xeon
// and an AMD EPYC (tests/golden/make_x86_tables.py checks all 2^32 inputs): T[0 .. 4095] holds rcp over [1, 2) by m >> 11,
// T[4096 .. 12287] rsqrt over [1, 2) and [2, 4) by m >> 11.  Zero / subnormal -> inf of that sign, inf -> 0, NaN quieted, rcp results below the normal range
// flushed to zero, rsqrt of a negative -> the default NaN.  The device's functions and the CPU checker's are compared over
// every input by digest (acf_hip_selftest_x86, tests/test_gpu_arith.py).
// ------------------------------------------------------------------------
// (selects, not branches: the callers are column loops whose lanes must not diverge; the table index is in range for every input)
__device__ __forceinline__ uint32_t x86_rcp_bits(uint32_t u, const uint32_t* __restrict__ T)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    const uint32_t t = T[m >> 11];
    const int re = int((t >> 23) & 0xffu) + 127 - int(e);
    uint32_t r = re <= 0 ? s : (s | (uint32_t(re) << 23) | (t & 0x7fffffu)); // (underflow: flushed to zero)
    r = e == 0 ? (s | 0x7f800000u) : r;                                      // zero, subnormals: inf
    r = e == 0xffu ? (m ? (u | 0x400000u) : s) : r;                          // NaN quieted; 1 / inf = 0 of the same sign
    return r;
}
__device__ __forceinline__ uint32_t x86_rsqrt_bits(uint32_t u, const uint32_t* __restrict__ T)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    const int ue = int(e) - 127, odd = ue & 1, half = (ue - odd) / 2;
    const uint32_t t = T[4096 + ((odd << 12) | int(m >> 11))];
    uint32_t r = (uint32_t(int((t >> 23) & 0xffu) - half) << 23) | (t & 0x7fffffu);
    r = s ? 0xffc00000u : r;                                                 // negative: the default NaN
    r = e == 0 ? (s | 0x7f800000u) : r;                                      // +-0, subnormals: inf of that sign
    r = e == 0xffu ? (m ? (u | 0x400000u) : (s ? 0xffc00000u : 0u)) : r;     // NaN quieted; inf -> 0; -inf -> the default NaN
    return r;
}
__device__ __forceinline__ float x86_rcp(float x, const uint32_t* __restrict__ T)
{
    return __uint_as_float(x86_rcp_bits(__float_as_uint(x), T));
}
__device__ __forceinline__ float x86_rsqrt(float x, const uint32_t* __restrict__ T)
{
    return __uint_as_float(x86_rsqrt_bits(__float_as_uint(x), T));
}
// position-mixed digests of both functions over first + i * stride, i < count (the CPU checker forms the same sums): out[0] rcp, out[1] rsqrt
__global__ void __launch_bounds__(256) k_x86_digest(const uint32_t* __restrict__ T, uint32_t first, unsigned long long count, uint32_t stride,
    unsigned long long* __restrict__ out)
{
    unsigned long long a = 0, b = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256)
    {
        const uint32_t u = first + uint32_t(i * stride);
        const unsigned long long k = ((unsigned long long)u * 0x9e3779b97f4a7c15ull) | 1ull;
        a += k * x86_rcp_bits(u, T);
        b += k * x86_rsqrt_bits(u, T);
    }
    atomicAdd(&out[0], a);
    atomicAdd(&out[1], b);
}

} // namespace acfhip

// The kernels by stage of the path (each part opens namespace acfhip itself); the order is the order of their dependencies.
#include "kernels_input.hip.h"      // rgbConvert, u8 ingest, the apps' resize, plain convTri1
#include "kernels_channels.hip.h"   // smoothing + fused consumers, gradMag, convTri, gradMagNorm / gradHist / addChn
#include "kernels_levels.hip.h"     // imResample, rank cells, the fused level kernels
#include "kernels_ldcf_strip.hip.h" // LDCF, LDS-tile resample, the strip march
#include "kernels_cascade.hip.h"    // acfDetect1: staged, LDS-tiled, tail, sort + box mapping
#include "kernels_output.hip.h"     // bbNms + prune, record export
