// Host-callable launchers of the wavefront kernels (defined in wavefront.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"
#include "ptr_abi.h"

namespace ptrk {

struct LaunchConfig {
    uint32_t traceGrid;      // blocks of kTraceBlock threads for extend / connect (grid-stride)
    uint32_t* spill;         // traversal stack spill area: (kTraversalStackDepth-kLdsStackLevels) * traceGrid*kTraceBlock words
    uint32_t* workCounters;  // [2] work-queue heads of k_extend / k_connect, zeroed by the host before each launch
    int refillBelow;         // persistent waves hand out new rays once fewer than this many lanes are traversing
    uint32_t feederChunk = 256;   // slots a wave claims per atomic on the work head
};

void launchGenerate(const RenderParams& rp, const PathPool& pool, hipStream_t stream);
// aliveOut (nullable): += number of live slots the launch traced (host termination check)
void launchExtend(const SceneView& sc, const PathPool& pool, const LaunchConfig& cfg, uint32_t* aliveOut, bool count, hipStream_t stream);
// words k_shade sets to zero for the launches that follow it (each may be null)
struct ShadeResets {
    uint32_t* extendHead;    // work head of the next k_extend
    uint32_t* connectHead;   // work head of this iteration's k_connect
    uint32_t* nextAlive;     // live-slot counter of the next k_extend
    uint32_t drained;        // nonzero once most slots are dead: waves look at the state word alone before loading the rest
};
void launchShade(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const ShadeResets& resets, bool count,
                 hipStream_t stream);
void launchConnect(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const LaunchConfig& cfg, bool count,
                   hipStream_t stream);
// End of the frame: every busy slot of `pool` (the WHOLE pool) is run to the end of its path by one lane (k_tail_collect + k_tail_run).
// dList: pool.slots words; dListCount / dListHead: single words, zero on entry.
void launchTail(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const LaunchConfig& cfg, uint32_t* dList, uint32_t* dListCount,
                uint32_t* dListHead, bool count, hipStream_t stream);
// Adds outstanding light connections, reduces the slots of each pixel in fixed order and writes
// out[((localBand*PTR_BAND_ROWS + row) * width + x) * 3 + c] = sum / spp.
void launchResolve(const RenderParams& rp, const PathPool& pool, uint32_t partCount, float* dOut, hipStream_t stream);

// Multi-device gather: writes image[y][x][c] from the partitions' band buffers laid end to end (dPartOffset: float offsets).
void launchInterleaveBands(const float* dGathered, const uint64_t* dPartOffset, uint32_t parts, uint32_t width, uint32_t height, float* dImage,
                           hipStream_t stream);

void launchTraceRays(const SceneView& sc, const float4* dRays, uint64_t n, bool anyHit, PtrHit* dOut, const LaunchConfig& cfg,
                     uint64_t* dCounters, hipStream_t stream);

// First-hit feature buffers for every pixel of the frame (row 0 = top): albedo rgb|hit flag, encoded normal|distance.
void launchAovs(const RenderParams& rp, const SceneView& sc, uint32_t sample, float4* dAlbedo, float4* dNormal, const LaunchConfig& cfg,
                hipStream_t stream);

// Debug / known-answer kernels (tests only): evaluate and sample a material for a batch of inputs.
void launchDebugEvalBsdf(const float4* dMaterial, const RenderParams& rp, const float* dIn, uint64_t n, float* dOut,
                         hipStream_t stream);
void launchDebugSampleBsdf(const float4* dMaterial, const RenderParams& rp, const float* dIn, const uint32_t* dFront,
                           const uint32_t* dRng, uint64_t n, float* dOut, uint32_t* dRngOut, hipStream_t stream);
void launchDebugTexSample(const SceneView& sc, uint32_t texture, const float* dIn, uint64_t n, float4* dOut, hipStream_t stream);
// closest hit, surface record and next-ray origin per input ray: in n x 9 floats {origin, direction, next direction}, out n x 16 floats
// the material / feature set of the k_shade instantiation launchShade picks (bsdf.h kAllMaterials = the full kernel)
uint32_t shadeKernelSet(const RenderParams& rp, const SceneView& sc, bool count);
void launchDebugSurfaceHits(const SceneView& sc, const float* dIn, uint64_t n, float* dOut, const LaunchConfig& cfg, hipStream_t stream);
void launchDebugCameraRays(const RenderParams& rp, const uint32_t* dXys, uint64_t n, float* dOut, uint32_t* dRngOut,
                           hipStream_t stream);

}  // namespace ptrk
