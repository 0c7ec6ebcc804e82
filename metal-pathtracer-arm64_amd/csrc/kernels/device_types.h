// POD views shared by the host launcher and the HIP kernels: the HBM layout of a scene and of the
// wavefront path pool.  Everything is SoA / 16-byte vectors so one lane moves one dwordx4 per access.
#pragma once

#include <cstdint>
#include <hip/hip_runtime.h>

#include "bvh_layout.h"

namespace ptrk {

// Node (4 x float4 = 64 B):  n[0] = (c0.min, bits(ref0))  n[1] = (c0.max, bits(ref1))
//                            n[2] = (c1.min, 0)           n[3] = (c1.max, 0)
// Triangle (3 x float4 = 48 B, BVH leaf order):
//   t[0] = (v0, bits(materialIndex))  t[1] = (v0-v1, bits(kind<<30 | geomIndex))  t[2] = (v2-v0, bits(primIndex))
//   kind: 0 = mesh triangle (geomIndex = mesh index), 2 = rectangle half (geomIndex = rectangle index)
// Triangle normals (3 x float4, same order): world-space vertex normals (mesh) / rectangle normal.
// Sphere (float4 centre+radius; uint2 {sphereIndex, materialIndex}), BVH leaf order.
// Quantised node (2 x uint4 = 32 B): child boxes as 16-bit grid coordinates over the scene bounds
// (lo rounded down, hi rounded up, one extra cell of padding), so a node is TWO 16-byte fetches per lane:
//   q[0] = (c0.lo.x | c0.lo.y<<16, c0.lo.z | c0.hi.x<<16, c0.hi.y | c0.hi.z<<16, ref0)   q[1] = same for child 1
// The slab test runs in grid space: t = (q - (org - gridOrigin)/cell) * (cell / dir).
struct SceneView {
    const float4* nodes;
    const uint4* qnodes;
    float gridOrigin[3];
    float gridCell[3];
    float gridInvCell[3];          // 1 / gridCell
    uint32_t useQuantized;
    const float4* tris;
    const float4* triNormals;
    const float4* spheres;
    const uint2* sphereInfo;
    const float4* materials;       // kMaterialVec4 float4 per material
    const float4* rects;           // 5 float4 per rectangle (PtrRect layout)
    const float4* rectLights;      // kRectLightVec4 float4 per light: corner|area, edgeU|twoSided, edgeV|rectIndex, normal|has-triangles,
                                   // emission, then the light's own two triangles as the traversal stores them (v0, v0-v1, v2-v0 each)
    const int32_t* lightIndexByRect;
    const float4* envRgba;
    const float2* envCond;         // (threshold, bits(alias)) per texel
    const float2* envMarg;         // per row
    const float* envPdf;
    uint32_t rootRef;
    uint32_t materialCount;
    uint32_t rectCount;
    uint32_t rectLightCount;
    uint32_t envWidth;
    uint32_t envHeight;
    uint32_t envSampling;          // distribution present
    uint32_t nodeBytes;            // size of the node array in use (qnodes or nodes): buffer-descriptor range
    uint32_t triBytes;             // size of the triangle array
    uint32_t oversizeRef;          // leaf reference of the triangles kept out of the tree (kRefEmpty: none): every ray tests them first
    // PTR_WIDE_NODES=1 (experiment): 64 B four-wide quantised nodes for k_extend / k_connect.  wnodes[j] holds, for binary node j, the
    // 16 B child records of its grandchildren (a child that is a leaf keeps its own record; unused places are kRefEmpty), indexed
    // like the binary nodes so the child references stay valid; every second level of the binary tree is never visited.
    const uint4* wnodes;
    uint32_t wideBytes;
    uint32_t useWide;
    // ---- material textures (PTR_METAL_PBR only; all null / 0 when the scene has none) - kernels/texture.h
    const float4* triUv;           // 4 float4 per triangle, leaf order: (uv0, uv1) of the three vertices, then (uvPerWorld0, uvPerWorld1, 0, 0)
    const float4* triTangent;      // 3 float4 per triangle, leaf order: world-space vertex tangents, w = handedness (0: none)
    const float4* texels;          // every level of every texture, linear RGBA
    const uint4* texInfo;          // kTexInfoVec4 uint4 per texture
    const float4* materialTex;     // kMaterialTexVec4 float4 per material: texture transforms, indices, uv sets, pbr params
    uint32_t textureCount;
    uint32_t settleRectLights;     // every rectangle light has its two triangles on record and there are few enough of them: k_shade settles
                                   // specular connections itself (wavefront.hip, kind-3 records)
    uint32_t materialTypes;        // bit t set: some material of the scene has type t (k_shade instantiations for simple scenes)
    uint32_t stackLimit;           // traversal stack entries a ray of this scene can need at most (<= kTraversalStackDepth): LDS levels + spill levels
};

// per-material texture record (materialTex): [0..11] KHR_texture_transform rows of the six slots (base colour, metallic-roughness,
// normal, occlusion, emissive, transmission), [12] bits(texture index) of slots 0..3, [13] = bits(index slot 4, index slot 5,
// uv-set bits (bit k = slot k reads TEXCOORD_1), materialFlags), [14] pbrParams (metallic, roughness, occlusion strength, normal
// scale), [15] pbrExtras (alpha factor, alpha cutoff, transmission factor, alpha mode)
constexpr uint32_t kMaterialTexVec4 = 16u;

constexpr uint32_t kRectLightVec4 = 11u;

// Compact material record: the MaterialData fields the Embree-semantics integrator reads.
constexpr uint32_t kMaterialVec4 = 16u;
enum MaterialSlot : uint32_t {
    kMatBaseColorRoughness = 0,
    kMatTypeEta = 1,
    kMatEmission = 2,
    kMatConductorEta = 3,
    kMatConductorK = 4,
    kMatCoatParams = 5,
    kMatCoatTint = 6,          // w = pbr metallic
    kMatCoatAbsorption = 7,
    kMatCarpaintBase = 8,
    kMatCarpaintFlake = 9,
    kMatCarpaintBaseEta = 10,
    kMatCarpaintBaseK = 11,
    kMatDielectricSigmaA = 12,   // xyz absorption coefficient of a dielectric's interior (Metal media semantics)
    kMatSssSigmaA = 13,          // xyz sigma_a override, w > 0.5: the override is in force (Metal subsurface semantics)
    kMatSssSigmaS = 14,          // xyz sigma_s override, w anisotropy g
    kMatSssParams = 15,          // x mean free path, y method (>= 0.5: random walk), z coat enabled
};

struct CameraParams {
    float origin[3], lowerLeft[3], horizontal[3], vertical[3], u[3], v[3];
    float lensRadius;
};

// Exact n / d for a divisor fixed per render (Granlund & Montgomery, "Division by invariant integers using multiplication", fig. 4.1,
// N = 32): the two divisors of a camera ray - item -> (sample, local pixel) and pixel -> (x, y) - cost the generic 32-bit division's
// ~30 VALU instructions each in every k_shade visit; this is five.
struct DivU32 {
    uint32_t d, magic, shift1, shift2;
    __host__ __device__ __forceinline__ uint32_t quotient(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t t = __umulhi(magic, n);
#else
        const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(magic) * n) >> 32);
#endif
        return (t + ((n - t) >> shift1)) >> shift2;
    }
};
inline DivU32 makeDivU32(uint32_t d) {
    DivU32 r{d, 0u, 0u, 0u};
    if (d == 0u) return r;   // (never divided by: a render without pixels returns before any launch)
    uint32_t l = 0u;         // ceil(log2 d)
    while (l < 32u && (1ull << l) < d) ++l;
    r.magic = static_cast<uint32_t>(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
    r.shift1 = l < 1u ? l : 1u;
    r.shift2 = l > 1u ? l - 1u : 0u;
    return r;
}

struct RenderParams {
    CameraParams cam;
    uint32_t width, height;
    uint32_t maxDepth;
    uint32_t seedBase;
    uint32_t spp;                  // samples per pixel of THIS pass (all of them unless the frame is rendered in several passes)
    uint32_t sampleBase;           // index of the pass's first sample: sample s of the pass draws the stream of sample sampleBase + s
    uint32_t sppTotal;             // samples per pixel of the whole frame (the final division)
    uint32_t passFlags;            // bit 0: first pass (the output is overwritten), bit 1: last pass (divide by sppTotal)
    uint32_t itemCount;            // localPixels * spp; work item w = sample * localPixels + localPixel (one sample each)
    uint32_t localPixels;          // pixels owned by this partition
    DivU32 byWidth, byLocalPixels; // width and localPixels as divisors
    uint32_t mediaMode;            // PTR_METAL_* bits (0 = Embree-parity integrator)
    uint32_t sssMode;              // RenderSettings::SssMode, read only with PTR_METAL_SSS
    uint32_t sssMaxSteps;          // closest-hit queries per random walk (>= 1)
    uint32_t itemHeadFirst;        // items below this are pre-assigned to the slots by k_generate
    uint32_t itemsPerHead;         // the remaining items are split into kItemHeads ranges of this size (multiple of 64)
    uint32_t enableRussianRoulette, enableSpecularNee, enableMnee, enableMneeSecondary;
    uint32_t backgroundMode;
    float backgroundColor[3];
    float envRotation, envIntensity;
    // FireflyClampParams
    float clampFactor, clampFloor, throughputClamp, tailClampBase, tailClampRoughnessScale, minSpecularPdf, clampEnabled;
    float emissionScale;
    float clampMaxContribution, minSpecularPdfRaw;   // PTR_METAL_CLAMPS only
    float shadowSlack;             // test knob (PtrSettings.debugShadowSlack): 0 = the reference's shadow-ray length (quirk Q9)
};

// One pending light-connection ray of a path slot.
//   org = (origin, tmax)   dir = (direction, bits(kind))   a = (contribution or bsdf weight, 0)   b = (throughput, 0)
//   kind 0: any-hit; a is the finished contribution, zeroed by the connect kernel when occluded
//   kind 1: closest hit + rectangle-light evaluation; connect kernel replaces a by the contribution
//           (a = bsdf weight, w = bsdf pdf; b = throughput)
//   kind 2: MNEE second bounce (closest hit, delta scatter with a copy of the rng in org.w, then both of the above)
//   kind 3: any-hit up to tmax that ignores the two triangles of one rectangle (b.x = bits of their meta word): a specular connection
//           to a rectangle light whose distance and contribution k_shade has already worked out; zeroed when occluded
struct ShadowRecordView {
    float4* org;
    float4* dir;
    float4* a;
    float4* b;
};

// The unclaimed work items are split into kItemHeads contiguous ranges, each with its own head counter, so the
// per-wave reservations of k_shade spread over 64 addresses (same-address atomics retire at ~88/us chip-wide; with
// one-sample items a single head was good for 0.76 ms of every k_shade launch).
constexpr uint32_t kItemHeads = 64u;
constexpr uint32_t kItemHeadStride = 64u;   // words between two heads (256 B): one cache line / L2 channel each
constexpr uint32_t kItemHeadWords = kItemHeads * kItemHeadStride + 1u;   // + the "every range is dry" word

// Record slots per path: 0 rect-light NEE, 1 environment NEE, 2 specular-NEE environment,
// 3 specular-NEE rectangle lights, 4 MNEE two-bounce chain.
constexpr uint32_t kRecSlots = 5u;

struct PathPool {
    // Per-slot path state.  Every field a kernel does not need stays out of its loads: k_extend reads the two ray
    // words and writes 8 B; k_connect reads the records that are pending (through the connect list); k_shade streams
    // 72 B in and 64 B out per live slot (it was 96 / 80 with one 16 B state word, a 16 B hit and padded rays).
    float4* ray0;          // (origin.xyz, direction.x)
    float4* ray1;        // (direction.y, direction.z, pdf of the last BSDF sample, bits(flags))
    float2* hit;           // (t, bits(hit word)) - barycentrics are recomputed by k_shade from the same operands
    float4* thr;           // (throughput.xyz, bits(rng state))
    float4* accum;         // (radiance sum of the slot's current work item, bits(work item))
    uint32_t* flushItem;   // valid while kFlagFlush: the finished item whose sum is published once its last records have landed
    uint4* medium;         // media mode only: stack of up to 8 dielectric material ids (16 bit each), depth in the flags
    float2* cone;          // textured scenes only: ray cone of the slot's path (width at the ray origin, spread), shaders/pathtrace.metal:129-160
    uint32_t* signature;   // counting build only: per-slot path signature (see kSig* below); null otherwise
    float4* itemAccum;     // [itemCount] finished work items (summed per pixel, in sample order, by k_resolve); w = signature
    uint32_t* nextItem;    // [kItemHeadWords] head k at [k * kItemHeadStride]: next unclaimed item of range k (k_shade)
    ShadowRecordView rec[kRecSlots];
    uint2* itemReserve;        // [slots/64] per-wave reservation {next, end} of work items (one atomic per 64 items)
    const uint32_t* pixelOfLocal;  // local pixel -> y*width + x
    const float4* zero;        // one float4 of zeros: where k_shade points the loads of records that are not pending
    uint64_t* counters;        // kCounterSlots
    uint32_t slots;            // slots of this pool (or of this group of the pool)
    uint32_t recStride;        // slots of the WHOLE pool: distance between the fields / record slots of `rec`
    // Connect list: k_shade appends every slot that queued light-connection records (entry = slot | record mask << kConnectMaskShift)
    // and k_connect walks the list instead of probing every slot's pending byte.  kConnectQueues sub-lists, each with its own
    // counter kConnectCountStride words apart (one shared counter would be a same-address atomic per k_shade wave); wave w of
    // k_shade appends to sub-list w % kConnectQueues, which therefore never holds more than connectRegion entries.  The counters are
    // double-buffered by iteration parity so that k_shade can clear the set the next iteration appends to.
    uint32_t* connectList;     // null: k_connect probes the slots (two-rays-per-lane build, end-of-frame kernels)
    uint32_t* connectCount;    // counters of this iteration
    uint32_t* connectClear;    // counters of the next iteration (k_shade zeroes them)
    uint32_t connectRegion;    // entries per sub-list (of the connect list and of the busy lists)
    // Busy lists (end of the frame, same sub-list layout): once most slots of a group are dead, k_shade appends every slot that still
    // needs a visit - its path goes on (kBusyAliveBit), or records / a finished item are outstanding - and the next k_extend and
    // k_shade walk that list instead of the slots, so an iteration costs what its live paths cost.  Three counter sets in rotation:
    // k_shade reads one, fills the next and clears the third.
    const uint32_t* busyIn;        // list this iteration's k_extend and k_shade walk (null: they walk the slots)
    const uint32_t* busyCountIn;
    uint32_t* busyOut;             // list k_shade fills for the next iteration (null: none)
    uint32_t* busyCountOut;
    uint32_t* busyCountClear;
};
constexpr uint32_t kBusyAliveBit = 1u << 31;
constexpr uint32_t kConnectQueues = 64u;
constexpr uint32_t kConnectCountStride = 64u;    // words: 256 B between counters
constexpr uint32_t kConnectMaskShift = 27u;      // slot indices stay below 2^27 (the pool is capped at 64 Mi slots)

// flags word (ray1.w)
constexpr uint32_t kFlagAlive = 1u << 0;
constexpr uint32_t kFlagLastDelta = 1u << 1;
constexpr uint32_t kFlagFlush = 1u << 2;        // a finished item waits for its last records: publish it at the next visit
constexpr uint32_t kFlagWalk = 1u << 3;         // the slot's ray is a step of a subsurface random walk (state in record slot 4)
constexpr uint32_t kFlagMediumShift = 4u;       // 4 bits: depth of the medium stack (0..8)
constexpr uint32_t kMaxMediumStack = 8u;
constexpr uint32_t kFlagDepthShift = 8u;        // 9 bits
constexpr uint32_t kFlagSpecDepthShift = 17u;   // 9 bits
constexpr uint32_t kFlagFieldMask = 0x1FFu;     // path depth is limited to 511 bounces
constexpr uint32_t kFlagPendingShift = 26u;     // 5 bits: the records queued at the last visit (same as pending[])
constexpr uint32_t kFlagPendingMask = (1u << kRecSlots) - 1u;

// Path signature (counting build, 1 spp): what the deterministic-stream tests use to say WHY a pixel differs from the
// oracle's.  bits 0..15: bit d set when the rectangle-light sample taken at path vertex d contributed (it was evaluated,
// non-zero and found unoccluded); bits 16..31: hash chain over the primitives hit, vertex by vertex (a miss included).
constexpr uint32_t kSigNeeBits = 16u;
__host__ __device__ inline uint32_t sigHashStep(uint32_t h, uint32_t primType, uint32_t geomIndex, uint32_t primIndex) {
    uint32_t x = (h * 0x9e3779b1u) ^ (primType * 0x85ebca6bu) ^ (geomIndex * 0xc2b2ae35u) ^ primIndex;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    return x & 0xFFFFu;
}

// parts of a k_shade visit the counting build times (PTR_VERBOSE=steps)
enum ShadePart : uint32_t {
    kShadePartLoad = 0,       // state loads, records landing, publishing a finished item
    kShadePartMiss = 1,       // background + MIS for a ray that left the scene
    kShadePartSurface = 2,    // hit reconstruction (triangle, normals, barycentrics), material fetch
    kShadePartEmitter = 3,    // a light reached by a BSDF-sampled ray
    kShadePartLightNee = 4,   // rectangle-light sample: BSDF value, weight, the light's own triangles, record
    kShadePartEnvNee = 5,     // environment sample
    kShadePartBsdf = 6,       // BSDF sampling, specular connections, throughput, roulette, next ray
    kShadePartItem = 7,       // claiming a work item and generating its camera ray
    kShadePartStore = 8,      // state stores, connect / busy list appends
    kShadePartWalk = 9,       // subsurface random-walk step (Metal semantics)
    kShadeParts = 10,
};

enum CounterSlot : uint32_t {
    kCntExtendRays = 0,
    kCntExtendNodes = 1,
    kCntExtendPrims = 2,
    kCntShadowRays = 3,
    kCntShadowNodes = 4,
    kCntShadowPrims = 5,
    kCntShadedHits = 6,
    kCntTriangleHits = 7,
    kCntPrimaryRays = 8,
    kCntShadowEarlyExit = 9,
    kCntStackOverflow = 10,
    kCntExtendLeaves = 11,          // k_extend only, counting build: lane-utilisation bookkeeping (PTR_VERBOSE=steps)
    kCntExtendWaveNodeSteps = 12,   // 64 x node steps executed by waves
    kCntExtendWavePrimSteps = 13,   // 64 x primitive steps executed by waves
    kCntExtendRefillPasses = 14,    // 64 x refill passes
    kCntExtendRefillTicks = 15,     // 64 x clock64 ticks / 16 spent in refill passes
    kCntExtendWaveTicks = 16,       // 64 x clock64 ticks / 16 a wave spent in the kernel
    kCntExtendActiveLanes = 17,     // sum over vote iterations of the lanes holding a ray
    kCntExtendLeafLanes = 18,       // ... of which at a leaf
    kCntExtendVoteIterations = 19,  // 64 x vote iterations
    // k_shade, counting build (PTR_VERBOSE=steps): lanes that reach each stage of a visit, and 64 x the waves that ran a visit at all
    kCntShadeWaves = 24,
    kCntShadeAlive = 25,        // slots with a ray that was traced
    kCntShadeSurface = 26,      // ... that hit a surface
    kCntShadeEmitter = 27,      // ... which was a light (the path ends there)
    kCntShadeLightEval = 28,    // rectangle-light samples that reach the BSDF evaluation
    kCntShadeLightPretest = 29, // ... and the test against the light's own triangles
    kCntShadeLightStored = 30,  // ... and queue a shadow ray
    kCntShadeBsdfSample = 31,   // lanes that sample the BSDF
    kCntShadeNeedItem = 32,     // lanes that ask for a new work item
    // k_shade, counting build: clock ticks per part of a visit (ShadePart), lane-summed [40..49] and per wave [50..59]
    kCntShadeLaneTicks = 40,
    kCntShadeWaveTicks = 50,
    kCounterSlots = 64,
};

}  // namespace ptrk
