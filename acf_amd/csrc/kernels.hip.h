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

} // namespace acfhip

// The kernels by stage of the path (each part opens namespace acfhip itself); the order is the order of their dependencies.
#include "kernels_input.hip.h"      // rgbConvert, u8 ingest, the apps' resize, plain convTri1
#include "kernels_channels.hip.h"   // smoothing + fused consumers, gradMag, convTri, gradMagNorm / gradHist / addChn
#include "kernels_levels.hip.h"     // imResample, rank cells, the fused level kernels
#include "kernels_ldcf_strip.hip.h" // LDCF, LDS-tile resample, the strip march
#include "kernels_cascade.hip.h"    // acfDetect1: staged, LDS-tiled, tail, sort + box mapping
#include "kernels_output.hip.h"     // bbNms + prune, record export
