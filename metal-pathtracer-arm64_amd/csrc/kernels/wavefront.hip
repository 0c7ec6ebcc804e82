// Wavefront path tracer kernels for MI355X (gfx950).
//
// One bounce = three launches over a group of resident path slots (the pool is cut into groups that run concurrently
// on separate HIP streams, csrc/host/hip_backend.cpp):
//   k_extend   closest-hit BVH traversal for every live slot     (persistent waves; cache latency + VALU issue)
//   k_shade    hit reconstruction, emission/background, light sampling, BSDF sampling, next ray or a new work
//              item for the slot                                 (one thread per slot; HBM stream of the path state)
//   k_connect  the light-connection records k_shade attached to the slots: any-hit, or closest hit + rectangle-light
//              evaluation (specular NEE / MNEE)                  (persistent waves, like k_extend)
// A slot renders work items (one pixel sample, or a chunk of them); every item has its own accumulator and k_resolve
// sums a pixel's items in a fixed order — results do not depend on scheduling, pool size or partition.
//
// Integrator semantics and RNG consumption order follow the reference's Embree backend
// (src/headless/EmbreeHeadlessRenderer.mm:2573-3130); the Metal twin is
// shaders/pathtrace.metal trace_path_software 5717-7284 / pathtraceIntegrateKernel 9698-9816.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "bsdf.h"
#include "device_types.h"
#include "launch.h"
#include <type_traits>
#include "traverse.h"
#include "texture.h"
#include "vec.h"

namespace ptrk {

namespace {

constexpr float kEps = 1.0e-4f;                  // ray epsilon of the Embree path
constexpr float kSpecNeePdfFloor = 1.0e-4f;
constexpr float kSpecNeeInvPdfClamp = 1.0e4f;
constexpr float kMisMin = 1.0e-4f;
constexpr float kMisMax = 0.9999f;
#ifndef PTR_SHADE_BLOCK
#define PTR_SHADE_BLOCK 128
#endif
constexpr uint32_t kShadeBlock = PTR_SHADE_BLOCK;

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// METAL: the instantiation that carries the Metal-only material models (k_shade<.., SSS = true>, the debug kernels).  The default
// kernel is compiled without them: as run-time flags they cost it 70 more spilled registers (k_shade is held at 96 VGPRs).
template <bool METAL = true>
__device__ __forceinline__ ClampCfg clampCfg(const RenderParams& rp) {
    ClampCfg c;
    c.factor = rp.clampFactor;
    c.floorLum = rp.clampFloor;
    c.throughput = rp.throughputClamp;
    c.tailBase = rp.tailClampBase;
    c.tailRoughScale = rp.tailClampRoughnessScale;
    c.minSpecPdf = rp.minSpecularPdf;
    c.enabled = rp.clampEnabled >= 0.5f;
    c.thinDielectrics = (rp.mediaMode & PTR_METAL_THIN) != 0u;
    c.metalSpecular = (rp.mediaMode & PTR_METAL_SPECULAR) != 0u;
    c.metalSss = (rp.mediaMode & PTR_METAL_SSS) != 0u;
    c.sssMode = rp.sssMode;
    c.metalPbr = (rp.mediaMode & PTR_METAL_PBR) != 0u;
    c.metalClamps = METAL && (rp.mediaMode & PTR_METAL_CLAMPS) != 0u;
    c.maxContribution = METAL ? rp.clampMaxContribution : 0.0f;
    c.minSpecPdfRaw = METAL ? rp.minSpecularPdfRaw : 0.0f;
    return c;
}

// Next-event estimation: the Embree path counts a light sample when the BSDF has a density for its direction and weights it with the
// unclamped balance heuristic (E:2753-2756, 2800-2802); the Metal kernel (PTR_METAL_CLAMPS, pathtrace.metal:6532-6552, 6624-6645) counts
// it when the BSDF value is positive, clamps the weight to [1e-4, 0.9999] and uses 1 where the BSDF reports no density.
template <bool METAL>
__device__ __forceinline__ bool neeContributes(const BsdfEvalResult& be, const ClampCfg& cc) {
    if (be.isDelta) return false;
    if (METAL && cc.metalClamps) return smax(smax(be.value.x, be.value.y), be.value.z) > 0.0f;
    return be.pdf > 0.0f;
}
template <bool METAL>
__device__ __forceinline__ float neeWeight(float lightPdf, float bsdfPdf, const ClampCfg& cc) {
    if (!(METAL && cc.metalClamps)) return lightPdf / (lightPdf + bsdfPdf);
    float weight = 1.0f;
    if (bsdfPdf > 0.0f) {
        const float denom = lightPdf + bsdfPdf;
        if (denom > 0.0f) weight = clampf(lightPdf / denom, kMisMin, kMisMax);
    }
    return weight;
}

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint32_t laneId() { return __lane_id(); }

__device__ __forceinline__ uint32_t waveSum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void addCounter(uint64_t* counters, uint32_t slot, uint32_t value) {
    const uint32_t total = waveSum(value);
    if (laneId() == 0 && total != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(counters + slot), static_cast<unsigned long long>(total));
}

// Hands out work indices [0, total) to the idle lanes of a persistent wave.  The wave owns a chunk of
// `chunk` consecutive indices at a time (one global atomic per chunk); inside a chunk, idle lanes take
// consecutive indices by rank, so neighbouring lanes still load neighbouring path slots.
struct WaveFeeder {
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t* counter;
    uint32_t total, next, end, chunk;
    bool exhausted;
    // chunkSize: 256 while the pool is full; the host passes a larger one once most slots are dead (end of the frame),
    // because same-address atomics retire at ~88 per microsecond and a 16 Mi-slot pool in 256-slot chunks costs
    // 0.75 ms per launch in atomics alone, however little work is left.
    // Once the chunk is so large that one chunk per resident wave covers the whole list (the host does that when the
    // pool has drained), wave w simply owns chunk w and the head is never touched: a launch over a drained pool
    // otherwise pays one atomic per wave (8192 -> 93 us) to learn that nothing is left.  With a full pool the static
    // first chunk measured 8 % slower than claiming everything from the head, so it is not used there.
    __device__ __forceinline__ bool allStatic() const { return static_cast<uint64_t>((gridDim.x * blockDim.x) >> 6) * chunk >= total; }
    __device__ __forceinline__ void init(uint32_t* c, uint32_t n, uint32_t chunkSize) {
        counter = c;
        total = n;
        chunk = chunkSize;
        next = 0u;
        end = 0u;
        exhausted = (n == 0u);
        if (allStatic()) {
            // readfirstlane: the compiler cannot see that threadIdx.x >> 6 is the same for the whole wave
            const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
            const uint64_t first = static_cast<uint64_t>(wave) * chunkSize;
            exhausted = first >= n;
            next = exhausted ? 0u : static_cast<uint32_t>(first);
            end = exhausted ? 0u : static_cast<uint32_t>(min(first + chunkSize, static_cast<uint64_t>(n)));
        }
    }
    // must be called by all 64 lanes (converged).  Every idle lane gets an index while any are left: a chunk that
    // runs out in the middle of a pass is followed by the next one in the same pass.
    __device__ __forceinline__ uint32_t take(bool idle) {
        const unsigned long long mask = __ballot(idle);
        if (mask == 0ull || exhausted) return kNone;
        const uint32_t need = static_cast<uint32_t>(__popcll(mask));
        const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << laneId()) - 1ull)));
        uint32_t idx = kNone, served = 0u;
        while (served < need) {
            if (next >= end) {
                if (allStatic()) {
                    exhausted = true;
                    break;
                }
                uint32_t base = 0u;
                if (laneId() == 0u) base = atomicAdd(counter, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total) {
                    exhausted = true;
                    break;
                }
                next = base;
                end = min(base + chunk, total);
            }
            const uint32_t n = min(end - next, need - served);
            if (idle && rank >= served && rank < served + n) idx = next + (rank - served);
            next += n;
            served += n;
        }
        return idx;
    }
};

// One wave's view of the kConnectQueues sub-lists of a connect list or a busy list (PathPool): the counters become one dense index
// space (lane l keeps the number of entries in sub-lists 0..l-1) and an index is turned into a position in the list's storage with
// six cross-lane reads.
struct SubLists {
    uint32_t before;   // lane l: entries in sub-lists 0 .. l-1
    uint32_t total;
    __device__ __forceinline__ void init(const uint32_t* counts, uint32_t region) {
        const uint32_t mine = min(counts[laneId() * kConnectCountStride], region);
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (static_cast<int>(laneId()) >= off) incl += up;
        }
        before = incl - mine;
        total = __builtin_amdgcn_readlane(incl, 63);
    }
    // must be called by all 64 lanes (converged); idx < total, or anything for lanes that will not use the result
    __device__ __forceinline__ uint32_t position(uint32_t idx, uint32_t region) const {
        // largest sub-list whose first index is <= idx (empty sub-lists share their successor's start and lose)
        uint32_t queue = 0u;
#pragma unroll
        for (uint32_t step = kConnectQueues / 2u; step != 0u; step >>= 1) {
            const uint32_t first = __shfl(before, static_cast<int>(queue + step), 64);
            if (first <= idx) queue += step;
        }
        const uint32_t first = __shfl(before, static_cast<int>(queue), 64);
        return queue * region + (idx - first);
    }
};

// Busy lists (PathPool::busyIn) exist from the moment the work items run out; a launch walks the list instead of the slots once the
// list is short enough to pay: k_extend as soon as half the slots are off it (a list entry costs one more dependent load, a dead
// slot a wasted probe), k_shade only below a quarter (its slot state is 16 B words in nine arrays: out of slot order every word
// costs a 64 B sector, so the list only wins once fewer than ~150/576 of the slots are busy).
#ifndef PTR_BUSY_EXTEND_DIV
#define PTR_BUSY_EXTEND_DIV 2
#endif
#ifndef PTR_BUSY_SHADE_DIV
#define PTR_BUSY_SHADE_DIV 4
#endif
#ifndef PTR_LIGHT_PRETEST   // A/B switch of the light self-occlusion test in k_shade
#define PTR_LIGHT_PRETEST 1
#endif
constexpr uint32_t kItemReserve = 64u;   // work items a wave reserves per atomic on the global head

// ---------------------------------------------------------------- camera
__device__ __forceinline__ void cameraRay(const RenderParams& rp, uint32_t x, uint32_t y, uint32_t& rng, f3& org, f3& dir) {
    const CameraParams& c = rp.cam;
    const float u = (static_cast<float>(x) + rngNext(rng)) / static_cast<float>(rp.width);
    float v = (static_cast<float>(y) + rngNext(rng)) / static_cast<float>(rp.height);
    v = 1.0f - v;
    f3 d = ((ld3(c.lowerLeft) + u * ld3(c.horizontal)) + v * ld3(c.vertical)) - ld3(c.origin);
    f3 o = ld3(c.origin);
    if (c.lensRadius > 0.0f) {
        // thin lens: rejection-sample the unit disk, at most 8 pairs, (0,0) if all rejected
        float dx = 0.0f, dy = 0.0f;
        for (int i = 0; i < 8; ++i) {
            const float px = rngNext(rng) * 2.0f - 1.0f;
            const float py = rngNext(rng) * 2.0f - 1.0f;
            if (px * px + py * py <= 1.0f) {
                dx = px;
                dy = py;
                break;
            }
        }
        dx *= c.lensRadius;
        dy *= c.lensRadius;
        const f3 offset = ld3(c.u) * dx + ld3(c.v) * dy;
        o += offset;
        d -= offset;
    }
    org = o;
    dir = normalize(d);
}

// Start sample `s` of the slot's pixel: seed, primary ray, fresh path state.
__device__ __forceinline__ void beginSample(const RenderParams& rp, uint32_t pixel, uint32_t s, uint32_t& rng, f3& org, f3& dir) {
    rng = rngHash(rp.seedBase ^ pixel ^ ((rp.sampleBase + s) * 0x9e3779b9u));
    const uint32_t y = rp.byWidth.quotient(pixel);
    cameraRay(rp, pixel - y * rp.width, y, rng, org, dir);
}
// ... of work item `item` (= sample * localPixels + local pixel)
__device__ __forceinline__ void beginItem(const RenderParams& rp, const PathPool& pool, uint32_t item, uint32_t& rng, f3& org, f3& dir) {
    const uint32_t s = rp.byLocalPixels.quotient(item);
    beginSample(rp, pool.pixelOfLocal[item - s * rp.localPixels], s, rng, org, dir);
}

// make_primary_ray_cone (shaders/pathtrace.metal:141-152): the cone a textured scene's paths start with
__device__ __forceinline__ float2 primaryCone(const RenderParams& rp) {
    const CameraParams& cam = rp.cam;
    const float pixelX = length(ld3(cam.horizontal)) / smax(static_cast<float>(rp.width), 1.0f);
    const float pixelY = length(ld3(cam.vertical)) / smax(static_cast<float>(rp.height), 1.0f);
    const float pixelFootprint = smax(smax(pixelX, pixelY), 1.0e-6f);
    const f3 centre = (ld3(cam.lowerLeft) + 0.5f * ld3(cam.horizontal)) + 0.5f * ld3(cam.vertical);
    const float focus = length(centre - ld3(cam.origin));
    return make_float2(smax(2.0f * cam.lensRadius, 0.0f), pixelFootprint / smax(focus, 1.0e-6f));
}

// ---------------------------------------------------------------- environment
__device__ __forceinline__ f3 skyColor(f3 direction) {
    const f3 unit = normalize(direction);
    const float t = 0.5f * (unit.y + 1.0f);
    return mk3(1.0f) * (1.0f - t) + mk3(0.5f, 0.7f, 1.0f) * t;
}

__device__ __forceinline__ void envUv(f3 direction, float rotation, float& u, float& v) {
    const f3 unit = normalize(direction);
    const float c = cosf(rotation), s = sinf(rotation);
    const f3 r = mk3(unit.x * c - unit.z * s, unit.y, unit.x * s + unit.z * c);
    u = (atan2f(r.z, r.x) + kPi) / (2.0f * kPi);
    v = 0.5f - asinf(clampf(r.y, -1.0f, 1.0f)) / kPi;
}

// Bilinear lookup on level 0 with the -0.5 texel offset, wrap in x, clamp in y.
__device__ __forceinline__ f3 envLookup(const SceneView& sc, f3 direction, float rotation, float intensity) {
    if (sc.envWidth == 0u || sc.envHeight == 0u) return mk3(0.0f);
    float u, v;
    envUv(direction, rotation, u, v);
    const int W = static_cast<int>(sc.envWidth), H = static_cast<int>(sc.envHeight);
    const float fx = u * static_cast<float>(sc.envWidth) - 0.5f;
    const float fy = v * static_cast<float>(sc.envHeight) - 0.5f;
    int x0 = static_cast<int>(floorf(fx)), y0 = static_cast<int>(floorf(fy));
    int x1 = x0 + 1, y1 = y0 + 1;
    const float tx = fx - static_cast<float>(x0), ty = fy - static_cast<float>(y0);
    x0 %= W; if (x0 < 0) x0 += W;
    x1 %= W; if (x1 < 0) x1 += W;
    y0 = min(max(y0, 0), H - 1);
    y1 = min(max(y1, 0), H - 1);
    const f3 c00 = mk3(sc.envRgba[static_cast<size_t>(y0) * W + x0]);
    const f3 c10 = mk3(sc.envRgba[static_cast<size_t>(y0) * W + x1]);
    const f3 c01 = mk3(sc.envRgba[static_cast<size_t>(y1) * W + x0]);
    const f3 c11 = mk3(sc.envRgba[static_cast<size_t>(y1) * W + x1]);
    const f3 c0 = c00 * (1.0f - tx) + c10 * tx;
    const f3 c1 = c01 * (1.0f - tx) + c11 * tx;
    return (c0 * (1.0f - ty) + c1 * ty) * smax(intensity, 0.0f);
}

// Solid-angle pdf of the texel a direction looks up (note: offset by half a turn from the texel the
// sampler would have drawn it from — reference quirk Q2, kept).
__device__ __forceinline__ float envPdfOf(const SceneView& sc, f3 direction, float rotation) {
    if (!sc.envSampling) return 0.0f;
    float u, v;
    envUv(direction, rotation, u, v);
    u = clampf(u, 0.0f, 0.99999994f);
    v = clampf(v, 0.0f, 0.99999994f);
    const uint32_t x = min(static_cast<uint32_t>(u * static_cast<float>(sc.envWidth)), sc.envWidth - 1u);
    const uint32_t y = min(static_cast<uint32_t>(v * static_cast<float>(sc.envHeight)), sc.envHeight - 1u);
    const float value = sc.envPdf[static_cast<size_t>(y) * sc.envWidth + x];
    return (isfinite(value) && value > 0.0f) ? value : 0.0f;
}

__device__ __forceinline__ void envSample(const SceneView& sc, float uM, float uC, float uJ, float rotation, f3& dir, float& pdf) {
    uM = clampf(uM, 0.0f, 0.99999994f);
    uC = clampf(uC, 0.0f, 0.99999994f);
    uJ = clampf(uJ, 0.0f, 0.99999994f);
    const float rowChoice = uM * static_cast<float>(sc.envHeight);
    uint32_t row = min(static_cast<uint32_t>(rowChoice), sc.envHeight - 1u);
    const float2 me = sc.envMarg[row];
    if (rowChoice - static_cast<float>(row) >= me.x) row = min(__float_as_uint(me.y), sc.envHeight - 1u);
    const float colChoice = uC * static_cast<float>(sc.envWidth);
    uint32_t col = min(static_cast<uint32_t>(colChoice), sc.envWidth - 1u);
    const size_t rowOff = static_cast<size_t>(row) * sc.envWidth;
    const float2 ce = sc.envCond[rowOff + col];
    if (colChoice - static_cast<float>(col) >= ce.x) col = min(__float_as_uint(ce.y), sc.envWidth - 1u);
    const float jitterX = uC - floorf(uC);
    const float fx = (static_cast<float>(col) + jitterX) / static_cast<float>(sc.envWidth);
    const float fy = (static_cast<float>(row) + uJ) / static_cast<float>(sc.envHeight);
    const float theta = fy * kPi;
    const float phi = fx * (2.0f * kPi);
    const float st = sinf(theta), ct = cosf(theta);
    const f3 m = mk3(st * cosf(phi), ct, st * sinf(phi));
    const float cr = cosf(rotation), sr = sinf(rotation);
    dir = mk3(m.x * cr + m.z * sr, m.y, -m.x * sr + m.z * cr);
    pdf = sc.envPdf[rowOff + col];
}

// ---------------------------------------------------------------- surface reconstruction
struct Surface {
    f3 position, normal, hitShadingNormal;   // HitInfo fields of the oracle
    float t;
    uint32_t material;
    uint32_t primType;    // 0 mesh, 1 sphere, 2 rectangle
    uint32_t primIndex;   // rectangle index for primType 2
    uint32_t geomIndex;   // mesh index for primType 0
    uint32_t prim;        // leaf-order triangle index (primType 0 / 2)
    float bu, bv;         // barycentrics of a mesh-triangle hit
    bool frontFace, twoSided;
};

template <typename T>
__device__ __forceinline__ const T* byteOffset(const T* base, uint32_t bytes) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}

__device__ __forceinline__ Surface reconstruct(const SceneView& sc, f3 org, f3 dir, float t, uint32_t prim) {
    Surface s;
    s.t = t;
    s.prim = prim;
    s.bu = 0.0f;
    s.bv = 0.0f;
    s.position = org + t * dir;
    s.normal = mk3(0.0f, 1.0f, 0.0f);
    s.twoSided = false;
    if (prim & kHitSphereBit) {
        const uint32_t idx = prim & ~kHitSphereBit;
        const float4 sp = sc.spheres[idx];
        const uint2 info = sc.sphereInfo[idx];
        const f3 n = normalize(s.position - mk3(sp));
        s.normal = n;
        s.hitShadingNormal = n;
        s.frontFace = dot(dir, n) < 0.0f;
        s.twoSided = true;
        s.primType = 1u;
        s.geomIndex = 0u;
        s.primIndex = info.x;
        s.material = info.y;
        return s;
    }
    // (32-bit byte offsets from the arrays' bases - a primitive index has 26 bits - so that the loads take the scalar base + vector
    // offset form: the 64-bit multiply-add of a pointer + index costs four instruction slots and an address register pair per array)
    const uint32_t triOffset = prim * 48u;
    const float4* tp = byteOffset(sc.tris, triOffset);
    const float4* np = byteOffset(sc.triNormals, triOffset);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const float4 n0 = np[0], n1 = np[1], n2 = np[2];   // all six loads of the hit in flight together
    asm volatile("" ::"v"(a.x), "v"(b.x), "v"(c.x), "v"(n0.x), "v"(n1.x), "v"(n2.x));
    const f3 ng = cross(mk3(c), mk3(b));
    if (dot(ng, ng) > 0.0f) s.normal = normalize(ng);
    s.frontFace = dot(dir, s.normal) < 0.0f;
    const f3 adjusted = s.frontFace ? s.normal : -s.normal;
    f3 shading = adjusted;
    const uint32_t meta = __float_as_uint(b.w);
    s.material = __float_as_uint(a.w);
    s.geomIndex = 0u;
    if ((meta >> 30) == 0u) {
        s.primType = 0u;
        s.primIndex = __float_as_uint(c.w);
        s.geomIndex = meta & kTriGeomMask;
        float u, v;   // the traversal kernels store distance and primitive only
        triangleUv(mk3(a), mk3(b), mk3(c), org, dir, u, v);
        s.bu = u;
        s.bv = v;
        const float w = 1.0f - u - v;
        const f3 interp = (w * mk3(n0) + u * mk3(n1)) + v * mk3(n2);
        if (dot(interp, interp) > 0.0f) {
            shading = normalize(interp);
            if (dot(shading, adjusted) < 0.0f) shading = -shading;
        }
    } else {
        s.primType = 2u;
        s.primIndex = meta & kTriGeomMask;
        shading = mk3(n0);
        if (dot(shading, adjusted) < 0.0f) shading = -shading;
        s.twoSided = __float_as_uint(sc.rects[static_cast<size_t>(s.primIndex) * 5u + 4u].y) != 0u;
    }
    s.hitShadingNormal = shading;
    return s;
}

// Next-ray origin: pushed off the surface along the (hit record's) shading normal, to the side the
// direction leaves on, plus half an epsilon along the direction.
struct OffsetFrame {   // what offsetOrigin needs from the hit, whatever the direction: a visit offsets up to three rays from one hit
    f3 position, n;
    float distance;
};
__device__ __forceinline__ OffsetFrame offsetFrame(const Surface& s) {
    OffsetFrame of;
    f3 n = s.hitShadingNormal;
    if (dot(n, n) <= 0.0f) n = s.normal;
    if (dot(n, n) <= 0.0f) n = mk3(0.0f, 1.0f, 0.0f);
    of.n = normalize(n);
    of.distance = smax(fabsf(s.t) * 1.0e-4f, kEps);
    of.position = s.position;
    return of;
}
__device__ __forceinline__ f3 offsetOrigin(const OffsetFrame& of, f3 direction) {
    const float sign = dot(direction, of.n) >= 0.0f ? 1.0f : -1.0f;
    f3 o = of.position + of.n * (sign * of.distance);
    o += (direction * kEps) * 0.5f;
    return o;
}
__device__ __forceinline__ f3 offsetOrigin(const Surface& s, f3 direction) { return offsetOrigin(offsetFrame(s), direction); }

// Origin of the ray that leaves a separable-subsurface exit point (shaders/pathtrace.metal:6740-6766): offset_surface_point
// (:1210-1220), then 0.02 along the exit normal and 0.04 along the direction - the biases the reference uses to get clear
// of the mesh, since the exit point lies on the tangent plane, not on the surface.
__device__ __forceinline__ f3 sssExitOrigin(f3 exitPoint, f3 exitNormal, f3 direction) {
    const f3 n = (finite3(exitNormal) && dot(exitNormal, exitNormal) > 0.0f) ? normalize(exitNormal) : mk3(0.0f, 1.0f, 0.0f);
    const float sign = dot(direction, n) >= 0.0f ? 1.0f : -1.0f;
    f3 o = exitPoint + n * (sign * kEps * 4.0f);
    o += (direction * kEps) * 0.5f;
    o += n * smax(5.0e-3f * 4.0f, kEps * 32.0f);
    const f3 d = (finite3(direction) && dot(direction, direction) > 0.0f) ? normalize(direction) : n;
    o += d * smax(5.0e-3f * 8.0f, kEps * 32.0f);
    return o;
}

// pdf (solid angle, incl. 1/N light pick) of hitting rectangle `rectIndex` at `position` from `origin`
__device__ __forceinline__ float rectLightPdfForHit(const SceneView& sc, uint32_t primType, uint32_t rectIndex, f3 position, f3 origin) {
    if (sc.rectLightCount == 0u || sc.rectCount == 0u) return 0.0f;
    if (primType != 2u || rectIndex >= sc.rectCount) return 0.0f;
    if (sc.lightIndexByRect[rectIndex] < 0) return 0.0f;
    const float4* r = sc.rects + static_cast<size_t>(rectIndex) * 5u;
    const float area = length(cross(mk3(r[1]), mk3(r[2])));
    if (area <= 0.0f) return 0.0f;
    const f3 toLight = position - origin;
    const float distSq = dot(toLight, toLight);
    if (distSq <= 0.0f) return 0.0f;
    const float distance = sqrtf(distSq);
    const f3 direction = toLight / distance;
    float cosLight = dot(-direction, mk3(r[3]));
    if (__float_as_uint(r[4].y) != 0u) {
        cosLight = fabsf(cosLight);
    } else if (cosLight <= 0.0f) {
        return 0.0f;
    }
    if (cosLight <= 0.0f) return 0.0f;
    const float pdfArea = 1.0f / area;
    const float pdfDir = pdfArea * distSq / smax(cosLight, 1.0e-6f);
    return pdfDir * (1.0f / static_cast<float>(sc.rectLightCount));
}

// Emission + pdf of a rectangle light reached by a specular-chain ray (mnee_rect_light_hit twin).
__device__ __forceinline__ bool rectLightHit(const SceneView& sc, const Surface& s, f3 origin, float emissionScale, f3& emission, float& pdf) {
    if (sc.rectLightCount == 0u || sc.rectCount == 0u) return false;
    if (s.primType != 2u || s.primIndex >= sc.rectCount) return false;
    const int32_t li = sc.lightIndexByRect[s.primIndex];
    if (li < 0) return false;
    const float4* L = sc.rectLights + static_cast<size_t>(li) * kRectLightVec4;
    const bool twoSided = L[1].w != 0.0f;
    if (!s.frontFace && !twoSided) return false;
    emission = mk3(L[4]) * emissionScale;
    if (!(dot(emission, emission) > 0.0f)) return false;
    pdf = rectLightPdfForHit(sc, s.primType, s.primIndex, s.position, origin);
    return (pdf > 0.0f) && isfinite(pdf);
}

// ---- specular connections to rectangle lights, settled where they are made ----
// A delta bounce looks for a rectangle light straight along the sampled direction (specular NEE / MNEE, E:2856-2917): the reference
// traces a closest-hit ray and asks whether what it found is a light.  Which light that can be, and at what distance, follows from
// the lights' own rectangles alone: their two triangles ride in the light records exactly as the traversal stores them (rows 5..10),
// so the same test on the same operands gives the traversal's distance bit for bit.  A direction that meets no light's rectangle
// needs no ray at all (nearly every delta bounce of a glass object: config 4 queued one closest-hit ray per bounce and k_connect was
// its largest kernel); one that does becomes an any-hit query up to that distance which ignores the light's own two triangles
// (record kind 3), with the contribution already computed.
constexpr uint32_t kSettleLightsMax = 8u;   // scenes with more rectangle lights (or lights without triangles) keep the closest-hit record

__device__ __forceinline__ bool nearestRectLight(const SceneView& sc, f3 org, f3 dir, float& tHit, uint32_t& light, uint32_t& half) {
    float best = INFINITY;
    bool found = false;
    for (uint32_t li = 0; li < sc.rectLightCount; ++li) {
        const float4* T = sc.rectLights + static_cast<size_t>(li) * kRectLightVec4 + 5u;
#pragma unroll
        for (uint32_t h = 0; h < 2u; ++h) {
            float tt, tu, tv;
            if (triangleTest(mk3(T[h * 3u]), mk3(T[h * 3u + 1u]), mk3(T[h * 3u + 2u]), org, dir, kEps, best, tt, tu, tv)) {
                best = tt;
                light = li;
                half = h;
                found = true;
            }
        }
    }
    tHit = best;
    return found;
}

// what reconstruct() would return for a hit on half `half` of light `light` at distance t, as far as rectLightHit reads it
__device__ __forceinline__ Surface rectLightSurface(const SceneView& sc, f3 org, f3 dir, float t, uint32_t light, uint32_t half, uint32_t& metaWord) {
    const float4* T = sc.rectLights + static_cast<size_t>(light) * kRectLightVec4 + 5u + half * 3u;
    const float4 b = T[1], c = T[2];
    Surface s;
    s.t = t;
    s.prim = 0u;
    s.bu = 0.0f;
    s.bv = 0.0f;
    s.position = org + t * dir;
    s.normal = mk3(0.0f, 1.0f, 0.0f);
    const f3 ng = cross(mk3(c), mk3(b));
    if (dot(ng, ng) > 0.0f) s.normal = normalize(ng);
    s.hitShadingNormal = s.normal;
    s.frontFace = dot(dir, s.normal) < 0.0f;
    s.twoSided = false;
    metaWord = __float_as_uint(b.w);
    s.primType = 2u;
    s.primIndex = metaWord & kTriGeomMask;
    s.geomIndex = 0u;
    s.material = 0u;
    return s;
}

// Contribution of a specular-NEE ray whose closest hit is the light surface `ls` (kind 1 records once traced; kind 3 records up front).
__device__ __forceinline__ f3 rectContributionAt(const RenderParams& rp, const SceneView& sc, const ClampCfg& cc, const Surface& ls, f3 org, f3 weight,
                                                 float bsdfPdfIn, f3 thr) {
    f3 emission;
    float pdf;
    if (!rectLightHit(sc, ls, org, rp.emissionScale, emission, pdf)) return mk3(0.0f);
    const float lightPdf = smax(pdf, kSpecNeePdfFloor);
    const float invLightPdf = smin(1.0f / lightPdf, kSpecNeeInvPdfClamp);
    const float bsdfPdf = smax(bsdfPdfIn, kSpecNeePdfFloor);
    const float denom = lightPdf + bsdfPdf;
    float mis = denom > 0.0f ? (lightPdf / denom) : 0.0f;
    mis = clampf(mis, kMisMin, kMisMax);
    const f3 contrib = (weight * emission) * (mis * invLightPdf);
    return finite3(contrib) ? clampFirefly(thr, contrib, cc) : mk3(0.0f);
}

// An any-hit query ends with its hit word either still what it started as - kHitMiss, or for a kind-3 record the meta word of the
// rectangle it ignores (kind 2 in bits 31:30, the light material's shade key, never 0, in bits 29:26) - or replaced by a hit word:
// a triangle (bits 31:30 = 0) or a sphere (bit 31, no shade key).
__device__ __forceinline__ bool anyHitFound(uint32_t hitWord) {
    return (hitWord >> 30) == 0u || ((hitWord >> 30) == 2u && ((hitWord >> kHitKeyShift) & kHitKeyMask) == 0u);
}

struct PendingRay {
    f3 org, dir;
    float tmax;
};

__device__ __forceinline__ void storeRecord(const PathPool& pool, uint32_t slot, uint32_t which, uint32_t kind, f3 org, float tmaxOrBits,
                                            f3 dir, f3 a, float aw, f3 b) {
    const ShadowRecordView& r = pool.rec[which];
    r.org[slot] = mk4(org, tmaxOrBits);
    r.dir[slot] = mk4(dir, __uint_as_float(kind));
    r.a[slot] = mk4(a, aw);
    if (kind != 0u) r.b[slot] = mk4(b, 0.0f);
}

}  // namespace

// =====================================================================================================
// k_generate: first sample of every slot
// =====================================================================================================
__global__ void __launch_bounds__(256) k_generate(RenderParams rp, PathPool pool) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= pool.slots) return;
    // slot i starts with work item i; later items are claimed from pool.nextItem (initialised to `slots`)
    uint32_t flags = 0u, rng = 0u;
    f3 o = mk3(0.0f), d = mk3(0.0f);
    if (slot < rp.itemCount && rp.maxDepth > 0u) {
        beginItem(rp, pool, slot, rng, o, d);
        flags = kFlagAlive | kFlagLastDelta;
    }
    pool.ray0[slot] = mk4(o, d.x);
    pool.ray1[slot] = make_float4(d.y, d.z, 1.0f, __uint_as_float(flags));
    pool.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng));
    pool.accum[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(slot));
    if (pool.signature) pool.signature[slot] = 0u;
    if (pool.cone) pool.cone[slot] = primaryCone(rp);
}

// =====================================================================================================
// k_extend: closest hit for every live slot
// =====================================================================================================
// 8 waves/SIMD: the step loop with its repeated steps wants 66-68 VGPRs; holding it at 64 costs no spill in the loop
// and is 4 % faster than 7 waves
#ifndef PTR_TRAV_WAVES   // waves per SIMD the traversal kernels are compiled for (8: at most 64 VGPRs)
#define PTR_TRAV_WAVES 8
#endif
#define PTR_EXTEND_ATTR __attribute__((amdgpu_waves_per_eu(PTR_TRAV_WAVES, PTR_TRAV_WAVES)))
template <bool COUNT, bool ALIVE, int NODES>
__global__ void __launch_bounds__(kTraceBlock) PTR_EXTEND_ATTR k_extend(SceneView sc, PathPool pool, uint32_t* spill, uint32_t spillStride, uint32_t* workCounter,
                                                         int kRefillBelow, uint32_t feederChunk, uint32_t* aliveOut) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    stack.sp = 0u;
    TraceCounters cnt{0u, 0u};
    uint32_t rays = 0u;
    uint32_t aliveSeen = 0u;   // live slots this wave picked up (host termination check, end of the frame only)
    uint32_t refills = 0u;     // counting build: refill passes of this wave
    uint32_t refillCycles = 0u;   // counting build: clock64 ticks spent in them
    uint32_t activeLanes = 0u, leafLanes = 0u, voteIterations = 0u;   // counting build: occupancy of the vote iterations
    const long long kernelStart = COUNT ? clock64() : 0ll;

    SceneMem mem = sceneMem(sc);
    // end of the frame (the ALIVE instantiation): the work is the busy list k_shade left, not the slots (PathPool::busyIn)
    bool listed = ALIVE && pool.busyIn != nullptr;
    SubLists lists{0u, pool.slots};
    if (listed) {
        lists.init(pool.busyCountIn, pool.connectRegion);
        if (static_cast<uint64_t>(lists.total) * PTR_BUSY_EXTEND_DIV >= pool.slots) {   // still too full: walk the slots
            listed = false;
            lists.total = pool.slots;
        }
    }
    WaveFeeder feeder;
    // (a short list is dealt out in chunks of less than 256, down to one wave-load per resident wave: see k_connect)
#ifdef PTR_CONNECT_CHUNK_FIXED   // (A/B switch)
    const uint32_t listChunk = 256u;
#else
    const uint32_t listWaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t listChunk = min(256u, max(64u, ((lists.total + listWaves - 1u) / listWaves + 63u) & ~63u));
#endif
    feeder.init(workCounter, lists.total, listed ? listChunk : feederChunk);
    Trav t;
    t.cur = 0u;
    bool active = false;
    uint32_t mySlot = 0u;
    while (true) {
        const int nActive = __popcll(__ballot(active));
        if (nActive < kRefillBelow && !feeder.exhausted) {
            if (COUNT) ++refills;
            const long long refillStart = COUNT ? clock64() : 0ll;
            uint32_t idx = feeder.take(!active);
            if (listed) {
                const uint32_t entry = pool.busyIn[lists.position(idx != WaveFeeder::kNone ? idx : 0u, pool.connectRegion)];
                if (idx != WaveFeeder::kNone) idx = (entry & kBusyAliveBit) ? (entry & ~kBusyAliveBit) : WaveFeeder::kNone;
            }
            const uint32_t at = idx != WaveFeeder::kNone ? idx : 0u;
            const float4 r0 = pool.ray0[at], r1 = pool.ray1[at];   // both in flight before the liveness test
            const bool live = idx != WaveFeeder::kNone && (__float_as_uint(r1.w) & kFlagAlive);
            if (ALIVE) aliveSeen += static_cast<uint32_t>(__popcll(__ballot(live)));   // wave-uniform: stays in an SGPR
            if (live) {
                mySlot = idx;
                if (COUNT) ++rays;
                active = travBegin<NODES>(sc, t, mk3(r0), mk3(r0.w, r1.x, r1.y), kEps, INFINITY, false, stack);
                if (!active) pool.hit[idx] = make_float2(INFINITY, __uint_as_float(kHitMiss));
            }
            if (COUNT) {
                // make the loaded values "used" here so the pass is timed with its memory waits, as it runs
                asm volatile("" ::"v"(t.inv.x), "v"(t.oi.x));
                refillCycles += static_cast<uint32_t>(clock64() - refillStart);
            }
            continue;
        }
        if (nActive == 0) break;
        if (COUNT) {
            activeLanes += static_cast<uint32_t>(nActive);   // wave-uniform: per-lane copies, the wave sum is 64 x
            leafLanes += static_cast<uint32_t>(__popcll(__ballot(active && travAtLeaf(t))));
            ++voteIterations;
        }
        if (!travVote<COUNT, NODES>(sc, mem, t, active, stack, cnt)) {
            active = false;
            pool.hit[mySlot] = make_float2(t.hit.t, __uint_as_float(t.hit.prim));
        }
    }
    if (ALIVE) {
        // one atomic per persistent wave (8192 per launch).  Counting in k_shade took one per 64 slots: 262 k atomics on
        // one address for a 16 Mi-slot pool = 3 ms per launch, exactly when the frame is draining.
        if (laneId() == 0 && aliveSeen != 0u) atomicAdd(aliveOut, aliveSeen);
    }
    if (COUNT) {
        addCounter(pool.counters, kCntExtendRays, rays);
        addCounter(pool.counters, kCntExtendNodes, cnt.nodes);
        addCounter(pool.counters, kCntExtendPrims, cnt.prims);
        addCounter(pool.counters, kCntExtendLeaves, cnt.leaves);
        addCounter(pool.counters, kCntExtendWaveNodeSteps, cnt.waveNodeSteps);
        addCounter(pool.counters, kCntExtendWavePrimSteps, cnt.wavePrimSteps);
        addCounter(pool.counters, kCntExtendRefillPasses, refills);
        addCounter(pool.counters, kCntExtendActiveLanes, activeLanes >> 6);
        addCounter(pool.counters, kCntExtendLeafLanes, leafLanes >> 6);
        addCounter(pool.counters, kCntExtendVoteIterations, voteIterations);
        addCounter(pool.counters, kCntExtendRefillTicks, refillCycles >> 4);
        addCounter(pool.counters, kCntExtendWaveTicks, static_cast<uint32_t>((clock64() - kernelStart) >> 4));
    }
}

// ---- medium stack of the Metal media semantics: 8 dielectric material ids, 16 bits each, in one uint4 ----
__device__ __forceinline__ uint32_t mediumEntry(uint4 ms, uint32_t i) {
    const uint32_t w = i < 2u ? ms.x : (i < 4u ? ms.y : (i < 6u ? ms.z : ms.w));
    return (w >> ((i & 1u) * 16u)) & 0xFFFFu;
}

__device__ __forceinline__ uint4 mediumWithEntry(uint4 ms, uint32_t i, uint32_t id) {
    const uint32_t shift = (i & 1u) * 16u;
    const uint32_t keep = ~(0xFFFFu << shift), put = (id & 0xFFFFu) << shift;
    if (i < 2u) ms.x = (ms.x & keep) | put;
    else if (i < 4u) ms.y = (ms.y & keep) | put;
    else if (i < 6u) ms.z = (ms.z & keep) | put;
    else ms.w = (ms.w & keep) | put;
    return ms;
}

// ---- textured metallic-roughness material at a mesh-triangle hit (PTR_METAL_PBR; shaders/pathtrace.metal:5919-6400) ----
// Builds the per-hit material the Metal kernel writes back into its copy of MaterialData: base colour x texture, metallic /
// roughness x the G / B channels of the metallic-roughness texture, transmission, occlusion, emissive, the normal-mapped shading
// normal with the roughness widening of shortened normals, and the alpha test.  Level of detail from the path's ray cone;
// filtering rule in kernels/texture.h.  Returns true when the alpha test discards the hit (the ray passes through).
struct PbrHit {
    float4 ov[3];      // Mat overrides: base colour | roughness, emission, (metallic, transmission, occlusion, 0)
    f3 shadingNormal;
    bool twoSided;
};

__device__ __forceinline__ bool applyPbrTextures(const SceneView& sc, const Surface& sf, uint32_t materialIndex, const Mat& mat, f3 wo, float2 cone,
                                                 float hitDistance, uint32_t& rng, PbrHit& out) {
    const float4* mt = sc.materialTex + static_cast<size_t>(materialIndex) * kMaterialTexVec4;
    const float4 idx0 = mt[12], idx1 = mt[13], pbrParams = mt[14], pbrExtras = mt[15];
    const uint32_t texBase = __float_as_uint(idx0.x), texOrm = __float_as_uint(idx0.y), texNormal = __float_as_uint(idx0.z),
                   texOcc = __float_as_uint(idx0.w), texEmissive = __float_as_uint(idx1.x), texTrans = __float_as_uint(idx1.y);
    const uint32_t uvSets = __float_as_uint(idx1.z), materialFlags = __float_as_uint(idx1.w);
    // interpolate_uv / interpolate_tangent with saturated barycentric weights (:583-591, 640-739)
    f3 w = vmax0(mk3(1.0f - sf.bu - sf.bv, sf.bu, sf.bv));
    const float wsum = (w.x + w.y) + w.z;
    w = (wsum > 1.0e-8f) ? w / wsum : mk3(1.0f, 0.0f, 0.0f);
    const float4* tu = sc.triUv + static_cast<size_t>(sf.prim) * 4u;
    const float4 a = tu[0], b = tu[1], c = tu[2], per = tu[3];
    const float2 uv0 = make_float2((a.x * w.x + b.x * w.y) + c.x * w.z, (a.y * w.x + b.y * w.y) + c.y * w.z);
    const float2 uv1 = make_float2((a.z * w.x + b.z * w.y) + c.z * w.z, (a.w * w.x + b.w * w.y) + c.w * w.z);
    // footprint of the ray cone on the surface (:158-160, 178-185, 5947-5949)
    const float coneFootprint = smax(cone.x + cone.y * smax(hitDistance, 0.0f), 1.0e-7f);
    const float surfaceFootprint = coneFootprint / smax(fabsf(dot(normalize(sf.normal), normalize(wo))), 1.0e-3f);
    auto slot = [&](uint32_t k) { return texSlot(mt, k, (uvSets >> k) & 1u, uv0, uv1, per.x, per.y); };
    auto lodOf = [&](uint32_t tex, const TexSlot& t) { return texLod(sc, tex, t.uvPerWorld, surfaceFootprint); };
    const float4 one = make_float4(1.0f, 1.0f, 1.0f, 1.0f);

    const TexSlot sBase = slot(0u);
    const float4 baseSample = texSample(sc, texBase, sBase.u, sBase.v, lodOf(texBase, sBase), one);
    const float4 bcr = mat.p[kMatBaseColorRoughness];
    const f3 baseColor = mk3(bcr) * mk3(baseSample);
    float metallic = clampf(pbrParams.x, 0.0f, 1.0f), roughness = clampf(pbrParams.y, 0.0f, 1.0f);
    const bool disableOrm = (materialFlags & 1u) != 0u;   // kMaterialFlagDisableOrm
    if (!disableOrm && texOrm != kNoTexture && texOrm < sc.textureCount) {
        const TexSlot sOrm = slot(1u);
        const float4 mr = texSample(sc, texOrm, sOrm.u, sOrm.v, lodOf(texOrm, sOrm), one);
        metallic = clampf(mr.z * metallic, 0.0f, 1.0f);
        roughness = clampf(mr.y * roughness, 0.0f, 1.0f);
    }
    float transmission = clampf(pbrExtras.z, 0.0f, 1.0f);
    if (texTrans != kNoTexture && texTrans < sc.textureCount) {
        const TexSlot sT = slot(5u);
        transmission = clampf(transmission * texSample(sc, texTrans, sT.u, sT.v, lodOf(texTrans, sT), one).x, 0.0f, 1.0f);
    }
    transmission *= (1.0f - metallic);   // (the model applies this factor once more, as the reference does: :6194, 4666)
    // alpha test (:6196-6217): MASK compares with the cutoff, BLEND keeps the hit with probability alpha
    const float alpha = clampf(clampf(pbrExtras.x, 0.0f, 1.0f) * baseSample.w, 0.0f, 1.0f);
    if (pbrExtras.w > 0.5f) {
        const bool discard = pbrExtras.w < 1.5f ? (alpha < clampf(pbrExtras.y, 0.0f, 1.0f)) : (rngNext(rng) > alpha);
        if (discard) return true;
    }
    float occlusion = 1.0f;
    if (!disableOrm && texOcc != kNoTexture && texOcc < sc.textureCount) {
        const TexSlot sO = slot(3u);
        const float occ = texSample(sc, texOcc, sO.u, sO.v, lodOf(texOcc, sO), one).x;
        const float strength = clampf(pbrParams.z, 0.0f, 1.0f);
        occlusion = 1.0f + (occ - 1.0f) * strength;   // mix(1, occ, strength)
    }
    f3 emissive = mk3(mat.p[kMatEmission]);
    if (texEmissive != kNoTexture && texEmissive < sc.textureCount) {
        const TexSlot sE = slot(4u);
        emissive *= mk3(texSample(sc, texEmissive, sE.u, sE.v, lodOf(texEmissive, sE), one));
    }
    // normal map (:6281-6346): vertex tangent (Gram-Schmidt against the shading normal), else the triangle's UV derivatives, else an
    // arbitrary frame; the mapped normal is kept on the geometric normal's side
    f3 shadingNormal = sf.hitShadingNormal;
    if (dot(shadingNormal, shadingNormal) <= 0.0f) shadingNormal = sf.normal;
    shadingNormal = normalize(shadingNormal);
    const float normalScale = pbrParams.w;
    const bool useNormalMap = texNormal != kNoTexture && texNormal < sc.textureCount && normalScale > 1.0e-4f;
    if (useNormalMap) {
        const TexSlot sN = slot(2u);
        const float4 ns = texSample(sc, texNormal, sN.u, sN.v, lodOf(texNormal, sN), make_float4(0.5f, 0.5f, 1.0f, 1.0f));
        float normalLength = 1.0f;
        const f3 nts = decodeNormalMap(mk3(ns), normalScale, normalLength);
        f3 t = mk3(1.0f, 0.0f, 0.0f), bt = mk3(0.0f);
        bool hasBasis = false;
        if (sc.triTangent) {
            const float4* tt = sc.triTangent + static_cast<size_t>(sf.prim) * 3u;
            const float4 t0 = tt[0], t1 = tt[1], t2 = tt[2];
            f3 tw = (mk3(t0) * w.x + mk3(t1) * w.y) + mk3(t2) * w.z;
            const float tsign = (t0.w * w.x + t1.w * w.y) + t2.w * w.z;
            const float len2 = dot(tw, tw);
            tw = (finite3(tw) && len2 > 1.0e-12f) ? tw * (1.0f / sqrtf(len2)) : mk3(1.0f, 0.0f, 0.0f);
            if (fabsf(tsign) > 0.5f) {
                t = tw - shadingNormal * dot(shadingNormal, tw);
                if (finite3(t) && dot(t, t) > 1.0e-6f) {
                    t = normalize(t);
                    bt = normalize(cross(shadingNormal, t)) * (tsign < 0.0f ? -1.0f : 1.0f);
                    hasBasis = finite3(bt) && dot(bt, bt) > 1.0e-6f;
                }
            }
        }
        if (!hasBasis) {
            // compute_tangent_basis_from_uv (:843-911) in world space: the triangle's edges and the UV deltas of the normal map's set
            const float4* tp = sc.tris + static_cast<size_t>(sf.prim) * 3u;
            const f3 edge1 = -mk3(tp[1]), edge2 = mk3(tp[2]);   // stored as v0 - v1, v2 - v0
            const bool set1 = ((uvSets >> 2) & 1u) != 0u;
            const float2 q0 = set1 ? make_float2(a.z, a.w) : make_float2(a.x, a.y), q1 = set1 ? make_float2(b.z, b.w) : make_float2(b.x, b.y),
                         q2 = set1 ? make_float2(c.z, c.w) : make_float2(c.x, c.y);
            const float du1 = q1.x - q0.x, dv1 = q1.y - q0.y, du2 = q2.x - q0.x, dv2 = q2.y - q0.y;
            const float denom = du1 * dv2 - dv1 * du2;
            if (fabsf(denom) >= 1.0e-8f) {
                const float r = 1.0f / denom;
                f3 tangentW = (edge1 * dv2 - edge2 * dv1) * r;
                f3 bitangentW = (edge2 * du1 - edge1 * du2) * r;
                const float tl = dot(tangentW, tangentW), bl = dot(bitangentW, bitangentW);
                if (finite3(tangentW) && tl > 1.0e-12f && finite3(bitangentW) && bl > 1.0e-12f) {
                    tangentW = tangentW * (1.0f / sqrtf(tl));
                    bitangentW = bitangentW * (1.0f / sqrtf(bl));
                    t = tangentW - shadingNormal * dot(shadingNormal, tangentW);
                    if (finite3(t) && dot(t, t) > 1.0e-6f) {
                        t = normalize(t);
                        const float handed = dot(cross(shadingNormal, t), bitangentW) < 0.0f ? -1.0f : 1.0f;
                        bt = normalize(cross(shadingNormal, t)) * (handed * (per.z < 0.0f ? -1.0f : 1.0f));   // x sign of det(localToWorld), as the reference has it
                        hasBasis = true;
                    }
                }
            }
        }
        if (!hasBasis) {   // build_onb (:934-940)
            const f3 up = fabsf(shadingNormal.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
            t = normalize(cross(up, shadingNormal));
            bt = cross(shadingNormal, t);
        }
        f3 mapped = normalize((t * nts.x + bt * nts.y) + shadingNormal * nts.z);
        if (dot(mapped, sf.normal) < 0.0f) mapped = -mapped;
        shadingNormal = mapped;
        // shortened (filtered) normals widen the lobe (:6348-6388, without the first-hit gradient term)
        const float tok = smax((1.0f - normalLength) / smax(normalLength, 1.0e-6f), 0.0f);
        roughness = clampf(sqrtf(roughness * roughness + tok), 0.0f, 1.0f);
    }
    out.ov[0] = mk4(baseColor, roughness);
    out.ov[1] = mk4(emissive, 0.0f);
    out.ov[2] = make_float4(metallic, transmission, occlusion, 0.0f);
    out.shadingNormal = shadingNormal;
    out.twoSided = sf.twoSided || mat.p[kMatTypeEta].z > 0.5f;
    return false;
}

// =====================================================================================================
// k_shade
// =====================================================================================================
// The full kernel runs at 4 waves/SIMD (<= 128 VGPRs).  Round 1 ran it at 5 waves / 96 VGPRs with ~230 spilled registers in cold
// branches; with the slot body shared with the end-of-frame kernel the allocation at 96 spilled into the hot path (k_shade 372 -> 468 ms
// per frame), and measured on one box: 4 waves 1664, 5 waves 1570, 6 waves 1374 Msamples/s (profiles/r2_ab_shade_waves.txt).
#ifndef PTR_SHADE_WAVES
#define PTR_SHADE_WAVES 4
#endif
// Instantiations by material set.  The full kernel carries the registers of its heaviest branch (car paint, plastic, the metallic-
// roughness model) for every scene: 128 VGPRs with 29 spilled, 4 waves per SIMD.  A scene's materials are known at upload
// (SceneView::materialTypes), so launchShade picks the smallest compiled set that covers them; the lean sets need 94-96 registers
// without spills and run 5 waves per SIMD, which hides more of the kernel's dependent loads (config 2: k_shade -12 % per launch,
// frame +7.7 %; profiles/r3_ab_material_sets.txt).  6 waves (80 registers) spill into the hot path and lose 14 %.
constexpr uint32_t kDiffuseMaterials = (1u << 0) | (1u << 3) | (1u << 5);                 // Lambert, light, subsurface-as-Lambert
constexpr uint32_t kBasicMaterials = kDiffuseMaterials | (1u << 2);                       // ... and glass
constexpr uint32_t kMetalMaterials = kBasicMaterials | (1u << 1);                         // ... and metals
constexpr uint32_t kCarPaintMaterials = kBasicMaterials | (1u << 6);                      // car paint (+ the basic ones)
constexpr uint32_t kPbrMaterials = kBasicMaterials | (1u << 7) | kFeatureEnvironment;     // glTF: metallic-roughness under an environment map
// The light-connection records (device_types.h kRecSlots) a set's scenes can queue: 0 always; 1 needs an environment map; 2 and 3 a
// material that samples delta directions (smooth metal, glass, the metallic-roughness model), 2 the environment as well; 4 glass.
constexpr uint32_t shadeRecords(uint32_t mats) {
    const bool env = (mats & kFeatureEnvironment) != 0u;
    const bool delta = (mats & ((1u << 1) | (1u << 2) | (1u << 7))) != 0u;
    return 1u | (env ? 2u : 0u) | ((env && delta) ? 4u : 0u) | (delta ? 8u : 0u) | ((mats & (1u << 2)) ? 16u : 0u);
}
// waves per SIMD of each set's instantiation (512 / waves = its register budget).  Compiled without the SLP vectoriser (csrc/Makefile:
// its packed-fp32 pairs cost more moves and registers than they saved instructions) the diffuse and basic sets need 72 / 76 registers
// and run 6 waves, the metallic-roughness set 81 and the metal set 96 (5 waves).
#ifndef PTR_SHADE_WAVES_DIFFUSE
#define PTR_SHADE_WAVES_DIFFUSE 6
#endif
#ifndef PTR_SHADE_WAVES_BASIC
#define PTR_SHADE_WAVES_BASIC 6
#endif
#ifndef PTR_SHADE_WAVES_METAL
#define PTR_SHADE_WAVES_METAL 5
#endif
#ifndef PTR_SHADE_WAVES_PBR
#define PTR_SHADE_WAVES_PBR 5
#endif
// the instantiations with the Metal-only subsurface / PBR models (and textures) need 161-171 registers: at 4 waves they spilled ~150
// (3 waves: +20 % on the Metal variants of configs 3 and 4, +8 % on config 5's; 2 waves: as slow as 4; profiles/r3_ab_metal_waves.txt)
#ifndef PTR_SHADE_WAVES_SSS
#define PTR_SHADE_WAVES_SSS 3
#endif
// (the counting build - stage counters, per-part clocks, path signatures - gets 256 registers: it is a diagnostic, and with its extra
// state at 128 the allocator spilled around divergent regions)
constexpr int shadeWaves(bool count, bool sss, uint32_t mats) {
    return count ? 2
           : mats == kDiffuseMaterials ? PTR_SHADE_WAVES_DIFFUSE
           : mats == kBasicMaterials   ? PTR_SHADE_WAVES_BASIC
           : mats == kMetalMaterials   ? PTR_SHADE_WAVES_METAL
           : mats == kPbrMaterials     ? PTR_SHADE_WAVES_PBR
                                       : (sss ? PTR_SHADE_WAVES_SSS : PTR_SHADE_WAVES);
}
#define PTR_SHADE_WAVES_ATTR_M __attribute__((amdgpu_waves_per_eu(shadeWaves(COUNT, SSS, MATS), shadeWaves(COUNT, SSS, MATS))))
// Work items for the lanes of a converged wave whose lane l holds slot 64 w + l (dense k_shade).
// Each wave holds a reservation of kItemReserve consecutive items in HBM and refills it with ONE atomic on one of kItemHeads range
// heads (a per-lane or even per-wave-per-bounce atomic on one address caps at ~88 ops/us chip-wide; so does one shared head once
// items are single samples).  Returns the lane's item, or rp.itemCount when it wanted none / none is left.  Must be called by all
// 64 lanes.
__device__ __forceinline__ uint32_t claimItems(const RenderParams& rp, const PathPool& pool, const uint32_t slot, const bool needItem, uint2 res) {
    // res: the wave's reservation, pool.itemReserve[slot / 64] - requested at the top of the visit with the slot's state, because only
    // this wave position ever changes it (a load here would be one more trip to memory that nothing overlaps)
    const unsigned long long mask = __ballot(needItem);
    if (mask == 0ull) return rp.itemCount;
    const uint32_t waveId = slot / 64u;     // slot >= pool.slots lanes never need items
    const uint32_t firstLane = static_cast<uint32_t>(__ffsll(static_cast<long long>(mask))) - 1u;
    const uint32_t resOld = res.x;
    const uint32_t n = static_cast<uint32_t>(__popcll(mask));
    const uint32_t avail = res.y - res.x;
    uint32_t newBase = rp.itemCount, newEnd = rp.itemCount;   // "no items": ids >= itemCount are rejected by the caller
    if (avail < n) {
        // The wave keeps drawing from the range of its last reservation (initially range waveId % kItemHeads)
        // and moves on to the next range when that one is exhausted; a wave that finds all of them exhausted
        // raises the "dry" word so nobody else has to go round again.  Heads sit kItemHeadStride words apart:
        // same-LINE atomics serialise just like same-address ones.
        const uint32_t wid = __builtin_amdgcn_readfirstlane(waveId);
        uint32_t head = (res.y > rp.itemHeadFirst && rp.itemsPerHead > 0u) ? (res.y - 1u - rp.itemHeadFirst) / rp.itemsPerHead : wid % kItemHeads;
        head = min(head, kItemHeads - 1u);
        uint32_t* const dry = pool.nextItem + kItemHeads * kItemHeadStride;
        bool found = false;
        if (rp.itemsPerHead > 0u && __builtin_amdgcn_readfirstlane(*dry) == 0u) {
            for (uint32_t tries = 0; tries < kItemHeads && !found; ++tries) {
                const uint32_t headEnd = min(rp.itemHeadFirst + (head + 1u) * rp.itemsPerHead, rp.itemCount);
                uint32_t* const counter = pool.nextItem + head * kItemHeadStride;
                uint32_t base = headEnd;
                if (laneId() == firstLane && *counter < headEnd) base = atomicAdd(counter, kItemReserve);
                base = __shfl(base, static_cast<int>(firstLane), 64);
                if (base < headEnd) {
                    newBase = base;
                    newEnd = min(base + kItemReserve, headEnd);
                    found = true;
                } else {
                    head = (head + 1u) % kItemHeads;
                }
            }
            if (!found && laneId() == firstLane) *dry = 1u;
        }
    }
    const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << laneId()) - 1ull)));
    // lanes beyond the old reservation and the fresh one get no item (id = itemCount): only when every range is
    // exhausted, or for the last few items of the frame (ranges are multiples of kItemReserve, only the final
    // one is clipped)
    const uint32_t fresh = newEnd - newBase;   // 0 when every range is exhausted
    const uint32_t claimed = (rank < avail) ? (resOld + rank) : ((rank - avail < fresh) ? newBase + (rank - avail) : rp.itemCount);
    if (avail < n) {
        res.x = newBase + min(n - avail, fresh);
        res.y = newEnd;
    } else {
        res.x += n;
    }
    if (laneId() == firstLane) pool.itemReserve[waveId] = res;
    return needItem ? claimed : rp.itemCount;
}

// SSS: the instantiation with the Metal subsurface semantics (launched when PTR_METAL_SSS is set)
struct ShadeCounts {
    uint32_t shadedHit = 0u, triHit = 0u, primary = 0u;   // counting build
    uint32_t settled = 0u;   // specular connections settled without the reference's closest-hit ray (still booked as extend rays: the counters mirror the reference's)
    uint32_t stage[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // kCntShadeWaves ... kCntShadeNeedItem
    // clock ticks per part of a visit (kShadePart*): summed over the lanes that ran the part / booked once per wave that ran it
    uint32_t laneTicks[kShadeParts] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t waveTicks[kShadeParts] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
};

// counting build: a part of a visit runs between partBegin and partEnd inside ONE branch region, so every lane that reaches
// partEnd ran the whole part with the wave; the first of them books the wave's time
template <bool COUNT>
__device__ __forceinline__ long long partBegin() { return COUNT ? clock64() : 0ll; }
template <bool COUNT>
__device__ __forceinline__ void partEnd(ShadeCounts& counts, uint32_t part, long long t0) {
    if (COUNT) {
        const uint32_t ticks = static_cast<uint32_t>(clock64() - t0);
        counts.laneTicks[part] += ticks;
        const unsigned long long here = __ballot(true);
        if (laneId() + 1u == static_cast<uint32_t>(__ffsll(static_cast<long long>(here)))) counts.waveTicks[part] += ticks;
    }
}

// A wave-uniform read of scene data: through the scalar cache into scalar registers (the constant address space is what makes the
// compiler choose s_load; the scene arrays are not written while a render runs).
typedef float NativeFloat4 __attribute__((ext_vector_type(4)));
struct UniformRows {
    const __attribute__((address_space(4))) NativeFloat4* p;
    __device__ __forceinline__ float4 operator[](uint32_t i) const {
        const NativeFloat4 v = p[i];
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ UniformRows uniformRows(const float4* p) {
    return UniformRows{reinterpret_cast<const __attribute__((address_space(4))) NativeFloat4*>(reinterpret_cast<uintptr_t>(p))};
}

// Rectangle-light next-event estimation at a non-delta hit: picks a light and a point on it, weighs the sample against the BSDF and
// queues the shadow ray (record 0).  3 random numbers, drawn even if the sample is rejected.  Returns whether a ray was queued.
// ONE: the scene has a single rectangle light (the pick still draws its random number).
template <bool COUNT, bool SSS, uint32_t MATS, bool ONE>
__device__ __forceinline__ bool rectLightNee(const RenderParams& rp, const SceneView& sc, const PathPool& pool, uint32_t slot, const Mat& mat,
                                             const Surface& sf, const OffsetFrame& of, f3 n, f3 wo, f3 thr, const ClampCfg& cc, uint32_t depth,
                                             uint32_t& rng, ShadeCounts& counts) {
    const uint32_t nL = ONE ? 1u : sc.rectLightCount;
    const uint32_t sel = min(static_cast<uint32_t>(rngNext(rng) * static_cast<float>(nL)), nL - 1u);
    const float lu = rngNext(rng);
    const float lv = rngNext(rng);
    const float4* Lv = sc.rectLights + static_cast<size_t>(sel) * kRectLightVec4;
    const UniformRows Lu = uniformRows(sc.rectLights);
    auto row = [&](uint32_t i) { return ONE ? Lu[i] : Lv[i]; };
    const float4 l0 = row(0), l1 = row(1), l2 = row(2), l3 = row(3);
    const f3 samplePoint = (mk3(l0) + lu * mk3(l1)) + lv * mk3(l2);
    const f3 toLight = samplePoint - sf.position;
    const float distSq = dot(toLight, toLight);
    if (!(distSq > 0.0f && l0.w > 0.0f)) return false;
    const float distance = sqrtf(distSq);
    const f3 ldir = toLight / distance;
    float cosLight = dot(-ldir, mk3(l3));
    if (l1.w != 0.0f) cosLight = fabsf(cosLight);
    if (!(cosLight > 0.0f)) return false;
    const float pdfArea = 1.0f / l0.w;
    const float pdfDir = pdfArea * distSq / smax(cosLight, 1.0e-6f);
    const float pdf = pdfDir * (1.0f / static_cast<float>(nL));
    const f3 emission = mk3(row(4)) * rp.emissionScale;
    const float nDotL = smax(dot(n, ldir), 0.0f);
    if (!(pdf > 0.0f && isfinite(pdf) && (dot(emission, emission) > 0.0f) && nDotL > 0.0f)) return false;
    if (COUNT) counts.stage[4] += 1u;
    const BsdfEvalResult be = evalBsdf<SSS, MATS>(mat, sf.position, n, wo, ldir, cc);
    if (!neeContributes<SSS>(be, cc)) return false;
    const float w = neeWeight<SSS>(pdf, be.pdf, cc);
    f3 contrib = (emission * be.value) * nDotL;
    contrib *= w / pdf;
    if (!finite3(contrib)) return false;
    const f3 clamped = clampFirefly(thr, contrib, cc);
    if (!(clamped.x > 0.0f || clamped.y > 0.0f || clamped.z > 0.0f)) return false;
    // tfar is measured from the un-offset hit point (reference quirk Q9)
    // (rp.shadowSlack is a test knob, 0 in every product render: x * 1.0f is x)
    const float shadowMax = smax(distance * (1.0f - rp.shadowSlack) - kEps, kEps);
    const f3 shadowOrg = offsetOrigin(of, ldir);
    // The shadow ray starts off the surface but its length is measured from the surface (quirk Q9), so from any surface that faces the
    // light's plane it reaches the light's OWN rectangle and is occluded by it - after walking the whole scene on the way.  An any-hit
    // query is occluded as soon as one primitive is hit: test the light's two triangles first (the same test on the same operands as
    // the traversal's), and queue a shadow ray only when they do not settle it.  Half of config 2's shadow rays end here.
    // The light's own two triangles ride in its record (rows 5..10, l3.w says so); they are read here, where few registers are live,
    // not with the rest of the record.
    float tt, tu, tv;
    bool occludedByLight = false;
    if (COUNT) counts.stage[5] += 1u;
    if (PTR_LIGHT_PRETEST != 0 && l3.w != 0.0f) {
        occludedByLight = triangleTest(mk3(row(5)), mk3(row(6)), mk3(row(7)), shadowOrg, ldir, kEps, shadowMax, tt, tu, tv) ||
                          triangleTest(mk3(row(8)), mk3(row(9)), mk3(row(10)), shadowOrg, ldir, kEps, shadowMax, tt, tu, tv);
    }
    if (occludedByLight) return false;
    // a.w: depth of this vertex, for the path signature of the counting build
    if (COUNT) counts.stage[6] += 1u;
    storeRecord(pool, slot, 0u, 0u, shadowOrg, shadowMax, ldir, clamped, static_cast<float>(depth), mk3(0.0f));
    return true;
}

// One visit of a path slot: what k_shade does for its thread's slot.  MODE kShadeDense: lane l of wave w holds slot 64 w + l.
// kShadeListed: the lanes hold the slots of a busy list (end of the frame) - the wave is converged but its slots are arbitrary, so
// work items are claimed lane by lane.  kShadeTail: the caller is the end-of-frame kernel (k_tail_run), whose lanes hold arbitrary
// slots AND diverge - nothing in here may then rely on the wave (no ballots, no list appends).
// `listWave`: which sub-list this wave appends to (wave-uniform; unused in kShadeTail).
constexpr int kShadeDense = 0, kShadeListed = 1, kShadeTail = 2;
// TEX (only with SSS): the scene has material textures - the per-hit texture lookups and the path's ray cone are compiled in.  A
// separate instantiation because they cost registers whether or not a scene uses them: with the texture code in, the Metal-model
// kernel drops to 3 waves/SIMD and untextured Metal-semantics scenes ran 19-25 % slower than in round 1.
// MATS: the material types the scene can contain (bsdf.h kAllMaterials, or the set of a simple scene: see launchShade)
template <bool COUNT, bool SSS, bool TEX, int MODE, uint32_t MATS = kAllMaterials>
__device__ __forceinline__ void shadeSlot(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const uint32_t slot, const bool inRange,
                                          const bool drained, const uint32_t listWave, ShadeCounts& counts) {
    constexpr bool TAIL = MODE == kShadeTail;
    // Everything that depends only on the slot index is requested up front, and the record loads are pointed at a
    // zero word when the record is not pending, so the kernel has ~11 loads in flight per lane after ONE dependent
    // step (the state word) instead of walking a chain of ten load-wait pairs at 5 waves per SIMD.
    const long long tLoad = partBegin<COUNT>();
    const uint32_t at = inRange ? slot : 0u;
    float4 ray1v = pool.ray1[at];
    if (!TAIL && drained) {
        // end of the frame: most waves cover 64 dead slots, and the loads below would still stream 56 B per slot
        const uint32_t peek = __float_as_uint(ray1v.w);
        const bool busy = inRange && (peek & (kFlagAlive | kFlagFlush | (kFlagPendingMask << kFlagPendingShift))) != 0u;
        if (__ballot(busy) == 0ull) return;
    }
    const float4 ray0v = pool.ray0[at];
    uint2 reservation = make_uint2(0u, 0u);
    if (MODE == kShadeDense && (slot & ~63u) < pool.slots) reservation = pool.itemReserve[__builtin_amdgcn_readfirstlane(slot / 64u)];   // (wave-uniform)
    const float2 hitv = pool.hit[at];
    const float4 thr4 = pool.thr[at];
    const float4 acc4 = pool.accum[at];
    const uint32_t flagsIn = inRange ? __float_as_uint(ray1v.w) : 0u;
    const uint32_t pendingIn = (flagsIn >> kFlagPendingShift) & kFlagPendingMask;
    // (the records the instantiation's scenes never queue are not read: kRecords)
    constexpr uint32_t kRecords = SSS ? 0x1Fu : shadeRecords(MATS);
    float4 landed[kRecSlots];
#pragma unroll
    for (uint32_t k = 0; k < kRecSlots; ++k) {
        landed[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (kRecords & (1u << k)) {
            const float4* src = (pendingIn & (1u << k)) ? pool.rec[k].a + at : pool.zero;
            landed[k] = *src;
        }
    }
    asm volatile("" ::"v"(acc4.x), "v"(ray0v.x), "v"(hitv.x), "v"(thr4.x));   // keep the loads here: the compiler would sink each one next to its use
#pragma unroll
    for (uint32_t k = 0; k < kRecSlots; ++k) {
        if (kRecords & (1u << k)) asm volatile("" ::"v"(landed[k].x));
    }
    const bool active = inRange && (flagsIn & kFlagAlive);
    const bool touched = inRange && (active || pendingIn != 0u || (flagsIn & kFlagFlush));   // state/accum rewritten
    if (COUNT) {
        counts.stage[0] += 1u;   // every lane counts its wave: the sum is 64 x the waves
        counts.stage[1] += active ? 1u : 0u;
    }

    bool want[kRecSlots] = {false, false, false, false, false};
    bool stillAlive = false, needItem = false, walkFlag = false;
    uint32_t shadedHit = 0u, triHit = 0u, primary = 0u;
    uint32_t rng = __float_as_uint(thr4.w);
    uint32_t item = __float_as_uint(acc4.w);
    uint32_t depth = 0u, specDepth = 0u, mediumDepth = 0u;
    bool lastDelta = true, flushNext = false;
    f3 acc = mk3(0.0f), thr = mk3(1.0f), nextO = mk3(0.0f), nextD = mk3(0.0f);
    float lastPdf = 1.0f;
    uint32_t sig = 0u;   // counting build: path signature of the slot's current item
    float2 cone = make_float2(0.0f, 0.0f);   // textured scenes: ray cone of the path (width at the ray origin, spread)
    bool haveCone = false, newSample = false;

    // what the instantiation's scene cannot have is compiled out (kFeatureEnvironment, kFeatureMedia: see launchShade)
    const uint32_t envWidth = (MATS & kFeatureEnvironment) ? sc.envWidth : 0u;
    const bool envSampling = (MATS & kFeatureEnvironment) && sc.envSampling;
    const uint32_t mediaMode = (MATS & kFeatureMedia) ? rp.mediaMode : 0u;

    if (touched) {
        const ClampCfg cc = clampCfg<SSS>(rp);
        acc = mk3(acc4);
        // light connections queued last bounce have been resolved by k_connect: add them in slot order
#pragma unroll
        for (uint32_t k = 0; k < kRecSlots; ++k) {
            if ((kRecords & (1u << k)) && (pendingIn & (1u << k))) acc += mk3(landed[k]);
        }
        if (COUNT && pool.signature) {
            sig = pool.signature[slot];
            // a surviving rectangle-light sample carries the depth of its vertex in a.w (k_connect zeroes the whole record
            // when the ray is occluded)
            if ((pendingIn & 1u) && (landed[0].x > 0.0f || landed[0].y > 0.0f || landed[0].z > 0.0f)) {
                const uint32_t d = static_cast<uint32_t>(landed[0].w);
                if (d < kSigNeeBits) sig |= 1u << d;
            }
        }
        if (flagsIn & kFlagFlush) {
            // the previous work item of this slot is complete (its last connections just landed): publish it
            pool.itemAccum[pool.flushItem[slot]] = mk4(acc, __uint_as_float(sig));
            acc = mk3(0.0f);
            sig = 0u;
        }
        partEnd<COUNT>(counts, kShadePartLoad, tLoad);

        if (active) {
            depth = (flagsIn >> kFlagDepthShift) & kFlagFieldMask;
            specDepth = (flagsIn >> kFlagSpecDepthShift) & kFlagFieldMask;
            mediumDepth = (flagsIn >> kFlagMediumShift) & 0xFu;
            lastDelta = (flagsIn & kFlagLastDelta) != 0u;
            const f3 rayO = mk3(ray0v);
            const f3 rayD = mk3(ray0v.w, ray1v.x, ray1v.y);
            thr = mk3(thr4);
            lastPdf = ray1v.z;
            const uint32_t prim = __float_as_uint(hitv.y);
            if (COUNT && depth == 0u) primary = 1u;

            bool endPath = false;
            nextO = rayO;
            nextD = rayD;
            bool walking = false;   // SSS: the slot stays in (or enters) a subsurface random walk: no bounce bookkeeping this visit

            if (SSS && (flagsIn & kFlagWalk)) {
                const long long tWalk = partBegin<COUNT>();
                // ---- one step of a subsurface random walk: this ray was the walk's boundary query (bsdf.h: sssWalkStep) ----
                const ShadowRecordView& wr = pool.rec[4];
                const float4 w0 = wr.org[slot], w1 = wr.dir[slot], w2 = wr.a[slot], w3 = wr.b[slot];
                const uint32_t entryMaterial = __float_as_uint(w1.w);
                const Mat mat{sc.materials + static_cast<size_t>(min(entryMaterial, sc.materialCount - 1u)) * kMaterialVec4};
                SssWalk walk{rayO, rayD, mk3(w2), __float_as_uint(w2.w)};
                f3 hitPoint = rayO, outward = mk3(0.0f);
                if (prim != kHitMiss) {
                    const Surface bsf = reconstruct(sc, rayO, rayD, hitv.x, prim);
                    hitPoint = bsf.position;
                    outward = bsf.normal;   // the geometric normal as stored = the reference's face-forwarded normal turned back
                }
                BsdfSampleResult bs{mk3(0.0f), mk3(0.0f), 0.0f, false, 0, false, mk3(0.0f)};
                int outcome = sssWalkStep(mat, rp.sssMaxSteps, walk, prim != kHitMiss, hitv.x, hitPoint, outward, rng, bs);
                if (outcome == kWalkWalking) {
                    walking = true;
                    nextO = walk.position;
                    nextD = walk.direction;
                    wr.a[slot] = mk4(walk.throughput, __uint_as_float(walk.step));
                } else {
                    // the walk is over: its exit sample, or (abandoned) the ordinary sample at the entry point
                    const f3 entryN = mk3(w3);
                    if (outcome == kWalkFallback) bs = sampleBsdf<true>(mat, mk3(w0), entryN, entryN, -entryN, true, rng, cc);
                    if (bs.pdf <= 0.0f || dot(bs.dir, bs.dir) <= 0.0f || !finite3(bs.weight)) {
                        endPath = true;
                    } else {
                        specDepth = 0u;
                        thr *= bs.weight;
                        thr = clampThroughput(thr, cc);
                        const float maxComp = smax(smax(thr.x, thr.y), thr.z);
                        if (!finite3(thr) || maxComp <= 0.0f) {
                            endPath = true;
                        } else {
                            lastPdf = bs.pdf > 0.0f ? bs.pdf : lastPdf;
                            lastDelta = false;
                            if (outcome == kWalkSample) {
                                nextO = sssExitOrigin(bs.exitPoint, outward, bs.dir);
                            } else {
                                // offsetOrigin at the entry hit (its offset normal and distance were parked with the walk)
                                const f3 on = mk3(w1);
                                const float sign = dot(bs.dir, on) >= 0.0f ? 1.0f : -1.0f;
                                nextO = mk3(w0) + on * (sign * smax(fabsf(w0.w) * 1.0e-4f, kEps));
                                nextO += (bs.dir * kEps) * 0.5f;
                            }
                            nextD = bs.dir;
                            if (rp.enableRussianRoulette && depth >= 5u) {
                                const float p = clampf(maxComp, 0.05f, 0.95f);
                                if (rngNext(rng) > p) {
                                    endPath = true;
                                } else {
                                    thr /= p;
                                }
                            }
                        }
                    }
                }
                partEnd<COUNT>(counts, kShadePartWalk, tWalk);
            } else if (prim == kHitMiss) {
                // ---- escaped: background, MIS-weighted against environment sampling ----
                const long long tMiss = partBegin<COUNT>();
                f3 bg;
                if (rp.backgroundMode == PTR_BG_SOLID) {
                    bg = ld3(rp.backgroundColor);
                } else if (rp.backgroundMode == PTR_BG_ENVIRONMENT && envWidth > 0u) {
                    bg = envLookup(sc, rayD, rp.envRotation, rp.envIntensity);
                } else {
                    bg = skyColor(rayD);
                }
                float mis = 1.0f;
                const bool useMis = (!lastDelta) || rp.enableSpecularNee || rp.enableMnee;
                if (useMis && envSampling) {
                    const float lightPdf = envPdfOf(sc, rayD, rp.envRotation);
                    const float denom = lastPdf + lightPdf;
                    if (denom > 0.0f) mis = lastPdf / denom;
                    mis = clampf(mis, kMisMin, kMisMax);
                }
                acc += clampFirefly(thr, bg * mis, cc);
                if (COUNT) sig = (sig & 0xFFFFu) | (sigHashStep(sig >> 16, 7u, 0u, 0u) << 16);
                endPath = true;
                partEnd<COUNT>(counts, kShadePartMiss, tMiss);
            } else if (sc.materialCount == 0u) {
                endPath = true;
            } else {
                const long long tSurface = partBegin<COUNT>();
                const Surface sf = reconstruct(sc, rayO, rayD, hitv.x, prim);
                const OffsetFrame of = offsetFrame(sf);   // (the same operations offsetOrigin(sf, .) would repeat per ray)
                if (COUNT) {
                    shadedHit = 1u;
                    if (COUNT) counts.stage[2] += 1u;
                    triHit = sf.primType == 0u ? 1u : 0u;
                    sig = (sig & 0xFFFFu) | (sigHashStep(sig >> 16, sf.primType, sf.geomIndex, sf.primIndex) << 16);
                }
                if ((mediaMode & PTR_METAL_MEDIA) && mediumDepth > 0u) {
                    // Beer-Lambert over the segment just travelled inside the innermost medium (pathtrace.metal:5869-5876)
                    const uint32_t inside = mediumEntry(pool.medium[slot], mediumDepth - 1u);
                    const Mat mm{sc.materials + static_cast<size_t>(min(inside, sc.materialCount - 1u)) * kMaterialVec4};
                    const f3 sigma = mm.sigmaA();
                    if (sigma.x > 0.0f || sigma.y > 0.0f || sigma.z > 0.0f) {
                        const float segment = smax(hitv.x, 0.0f);
                        thr *= mk3(expf(-sigma.x * segment), expf(-sigma.y * segment), expf(-sigma.z * segment));
                    }
                }
                const uint32_t materialIndex = min(sf.material, sc.materialCount - 1u);
                Mat mat{byteOffset(sc.materials, materialIndex * (kMaterialVec4 * 16u))};
                const uint32_t type = mat.type();
                const f3 incident = normalize(rayD);
                const f3 wo = -incident;
                f3 n = sf.hitShadingNormal;
                if (dot(n, n) <= 0.0f) n = sf.normal;
                if (type == 2u) {                   // dielectrics shade with the geometric normal
                    n = sf.normal;                  // (as stored: the Embree backend does not turn it towards the ray)
                    if ((mediaMode & PTR_METAL_FACE_NORMAL) && !sf.frontFace) n = -n;   // set_face_normal, pathtrace.metal:1187-1191
                }
                n = normalize(n);
                bool hitTwoSided = sf.twoSided;
                bool passThrough = false;   // the alpha test of a textured material discarded the hit
                PbrHit pbrHit;
                if (SSS && cc.metalPbr && type == 7u) {
                    if (TEX && sf.primType == 0u && sc.textureCount > 0u && sc.triUv != nullptr) {
                        cone = pool.cone ? pool.cone[slot] : make_float2(0.0f, 0.0f);
                        haveCone = pool.cone != nullptr;
                        passThrough = applyPbrTextures(sc, sf, materialIndex, mat, wo, cone, hitv.x, rng, pbrHit);
                        if (!passThrough) {
                            mat.o = pbrHit.ov;
                            n = pbrHit.shadingNormal;
                            hitTwoSided = pbrHit.twoSided;
                        }
                    }
                    if (!passThrough) {
                        // an emissive metallic-roughness surface adds its emission and the path goes on (pathtrace.metal:6437-6442)
                        const f3 emission = mk3(mat.v(kMatEmission));
                        if ((emission.x != 0.0f || emission.y != 0.0f || emission.z != 0.0f) && (sf.frontFace || hitTwoSided)) {
                            acc += clampFirefly(thr, emission, cc);
                        }
                    }
                }

                partEnd<COUNT>(counts, kShadePartSurface, tSurface);
                if (SSS && passThrough) {
                    // pathtrace.metal:6206-6216: the ray carries on through the surface; counts as a specular bounce
                    nextO = offsetOrigin(of, rayD);
                    nextD = rayD;
                    lastPdf = 1.0f;
                    lastDelta = true;
                    specDepth += 1u;
                } else if (type == 3u) {
                    const long long tEmitter = partBegin<COUNT>();
                    if (COUNT) counts.stage[3] += 1u;
                    // ---- emitter reached by a BSDF-sampled ray ----
                    const float4 em = mat.v(kMatEmission);
                    f3 emission = mk3(em) * rp.emissionScale;
                    if (em.w > 0.0f && envWidth > 0u && sf.frontFace) {
                        emission *= envLookup(sc, -n, rp.envRotation, rp.envIntensity);
                    }
                    if ((dot(emission, emission) > 0.0f) && (sf.frontFace || sf.twoSided)) {
                        float mis = 1.0f;
                        const bool useMis = (!lastDelta) || rp.enableSpecularNee || rp.enableMnee;
                        if (useMis && sc.rectLightCount > 0u) {
                            const float lightPdf = rectLightPdfForHit(sc, sf.primType, sf.primIndex, sf.position, rayO);
                            const float denom = lastPdf + lightPdf;
                            if (denom > 0.0f) mis = lastPdf / denom;
                            mis = clampf(mis, kMisMin, kMisMax);
                        }
                        acc += clampFirefly(thr, emission * mis, cc);
                    }
                    endPath = true;
                    partEnd<COUNT>(counts, kShadePartEmitter, tEmitter);
                } else {
                    bool surfaceDelta = materialIsDelta(mat);
                    if (SSS && cc.metalPbr && type == 7u) surfaceDelta = mat.roughness01() <= 1.0e-3f;   // pathtrace.metal:4578-4581

                    // ---- rectangle-light NEE (3 random numbers, drawn even if the sample is rejected) ----
                    if (!surfaceDelta && sc.rectLightCount > 0u) {
                        const long long tLight = partBegin<COUNT>();
                        // one light (every Cornell-type scene): its record is the same for the whole wave and comes through scalar loads
                        const bool queued = sc.rectLightCount == 1u
                                                ? rectLightNee<COUNT, SSS, MATS, true>(rp, sc, pool, slot, mat, sf, of, n, wo, thr, cc, depth, rng, counts)
                                                : rectLightNee<COUNT, SSS, MATS, false>(rp, sc, pool, slot, mat, sf, of, n, wo, thr, cc, depth, rng, counts);
                        if (queued) want[0] = true;
                        partEnd<COUNT>(counts, kShadePartLightNee, tLight);
                    }

                    // ---- environment NEE (3 random numbers: marginal, conditional, jitter) ----
                    if (!surfaceDelta && envSampling) {
                        const long long tEnv = partBegin<COUNT>();
                        const float uM = rngNext(rng);
                        const float uC = rngNext(rng);
                        const float uJ = rngNext(rng);
                        f3 edir;
                        float epdf;
                        envSample(sc, uM, uC, uJ, rp.envRotation, edir, epdf);
                        const float nDotL = smax(dot(n, edir), 0.0f);
                        if (epdf > 0.0f && nDotL > 0.0f) {
                            const f3 envRadiance = envLookup(sc, edir, rp.envRotation, rp.envIntensity);
                            const BsdfEvalResult be = evalBsdf<SSS, MATS>(mat, sf.position, n, wo, edir, cc);
                            if (neeContributes<SSS>(be, cc)) {
                                const float w = neeWeight<SSS>(epdf, be.pdf, cc);
                                f3 contrib = (envRadiance * be.value) * nDotL;
                                contrib *= w / epdf;
                                if (finite3(contrib)) {
                                    const f3 clamped = clampFirefly(thr, contrib, cc);
                                    if (clamped.x > 0.0f || clamped.y > 0.0f || clamped.z > 0.0f) {
                                        storeRecord(pool, slot, 1u, 0u, offsetOrigin(of, edir), INFINITY, edir, clamped, 0.0f, mk3(0.0f));
                                        want[1] = true;
                                    }
                                }
                            }
                        }
                        partEnd<COUNT>(counts, kShadePartEnvNee, tEnv);
                    }

                    // ---- continue the path ----
                    const long long tBsdf = partBegin<COUNT>();
                    BsdfSampleResult bs{mk3(0.0f), mk3(0.0f), 0.0f, false, 0, false, mk3(0.0f)};
                    bool haveSample = false;
                    if (SSS && type == 5u && cc.metalSss && rp.sssMode == 2u && mat.v(kMatSssParams).y >= 0.5f && sf.frontFace) {
                        // random-walk subsurface scattering (pathtrace.metal:6650-6676): coat lobe, or into the medium
                        SssWalk walk;
                        const int outcome = sssWalkBegin(mat, sf.position, sf.normal, wo, incident, rng, cc, bs, walk);
                        haveSample = outcome == kWalkSample;
                        if (outcome == kWalkWalking) {
                            walking = true;
                            nextO = walk.position;
                            nextD = walk.direction;
                            // what the ordinary sample needs should the walk be abandoned: entry point and hit distance, the
                            // normal offsetOrigin pushes along, the material, the shading normal
                            f3 on = sf.hitShadingNormal;
                            if (dot(on, on) <= 0.0f) on = sf.normal;
                            if (dot(on, on) <= 0.0f) on = mk3(0.0f, 1.0f, 0.0f);
                            const ShadowRecordView& wr = pool.rec[4];
                            wr.org[slot] = mk4(sf.position, sf.t);
                            wr.dir[slot] = mk4(normalize(on), __uint_as_float(min(sf.material, sc.materialCount - 1u)));
                            wr.a[slot] = mk4(walk.throughput, __uint_as_float(0u));
                            wr.b[slot] = mk4(n, 0.0f);
                        }
                    }
                    if (COUNT) counts.stage[7] += 1u;
                    if (!SSS || (!haveSample && !walking)) bs = sampleBsdf<SSS, MATS>(mat, sf.position, n, wo, incident, sf.frontFace, rng, cc);
                    if (SSS && walking) {
                        // nothing else this visit: the walk's first boundary query is the slot's next ray
                    } else if (bs.pdf <= 0.0f || dot(bs.dir, bs.dir) <= 0.0f || !finite3(bs.weight)) {
                        endPath = true;
                    } else {
                        if ((mediaMode & PTR_METAL_MEDIA) && bs.mediumEvent != 0) {
                            // refraction into / out of a dielectric: push its material, or pop (pathtrace.metal:6694-6709)
                            if (bs.mediumEvent > 0) {
                                const uint32_t at = min(mediumDepth, kMaxMediumStack - 1u);   // a full stack overwrites its top
                                pool.medium[slot] = mediumWithEntry(pool.medium[slot], at, min(sf.material, sc.materialCount - 1u));
                                mediumDepth = min(mediumDepth + 1u, kMaxMediumStack);
                            } else if (mediumDepth > 0u) {
                                --mediumDepth;
                            }
                        }
                        const uint32_t nextSpecDepth = bs.isDelta ? (specDepth + 1u) : 0u;
                        specDepth = nextSpecDepth;
                        const bool dirValid = finite3(bs.dir);
                        const bool mneeEligible = rp.enableMnee && bs.isDelta && dirValid && type == 2u && nextSpecDepth == 1u;
                        const bool specNeeEligible = rp.enableSpecularNee && bs.isDelta && dirValid && !mneeEligible;

                        if (specNeeEligible || mneeEligible) {
                            // light reached straight along the specular direction
                            const f3 sdir = normalize(bs.dir);
                            const f3 sorg = offsetOrigin(of, sdir);
                            if (envSampling) {
                                const float envPdf = smax(envPdfOf(sc, sdir, rp.envRotation), kSpecNeePdfFloor);
                                const float invEnvPdf = smin(1.0f / envPdf, kSpecNeeInvPdfClamp);
                                const float bsdfPdf = smax(bs.pdf, kSpecNeePdfFloor);
                                const float denom = envPdf + bsdfPdf;
                                float mis = denom > 0.0f ? (envPdf / denom) : 0.0f;
                                mis = clampf(mis, kMisMin, kMisMax);
                                const f3 envColor = envLookup(sc, sdir, rp.envRotation, rp.envIntensity);
                                const f3 contrib = (bs.weight * envColor) * (mis * invEnvPdf);
                                if (finite3(contrib)) {
                                    const f3 clamped = clampFirefly(thr, contrib, cc);
                                    if (clamped.x > 0.0f || clamped.y > 0.0f || clamped.z > 0.0f) {
                                        storeRecord(pool, slot, 2u, 0u, sorg, INFINITY, sdir, clamped, 0.0f, mk3(0.0f));
                                        want[2] = true;
                                    }
                                }
                            }
                            if (sc.rectLightCount > 0u && sc.settleRectLights) {
                                if (COUNT) counts.settled += 1u;
                                float tl;
                                uint32_t li = 0u, half = 0u;
                                if (nearestRectLight(sc, sorg, sdir, tl, li, half)) {
                                    uint32_t ignore;
                                    const Surface ls = rectLightSurface(sc, sorg, sdir, tl, li, half, ignore);
                                    const f3 c = rectContributionAt(rp, sc, cc, ls, sorg, bs.weight, bs.pdf, thr);
                                    if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f) {
                                        storeRecord(pool, slot, 3u, 3u, sorg, tl, sdir, c, 0.0f, mk3(__uint_as_float(ignore), 0.0f, 0.0f));
                                        want[3] = true;
                                    }
                                }
                            } else if (sc.rectLightCount > 0u) {
                                storeRecord(pool, slot, 3u, 1u, sorg, INFINITY, sdir, bs.weight, bs.pdf, thr);
                                want[3] = true;
                            }
                        }
                        if (mneeEligible && rp.enableMneeSecondary) {
                            const f3 sdir = normalize(bs.dir);
                            storeRecord(pool, slot, 4u, 2u, offsetOrigin(of, sdir), __uint_as_float(rng), sdir, bs.weight, bs.pdf, thr);
                            want[4] = true;
                        }

                        thr *= bs.weight;
                        thr = clampThroughput(thr, cc);
                        const float maxComp = smax(smax(thr.x, thr.y), thr.z);
                        if (!finite3(thr) || maxComp <= 0.0f) {
                            endPath = true;
                        } else {
                            lastPdf = bs.pdf > 0.0f ? bs.pdf : lastPdf;
                            lastDelta = bs.isDelta;
                            nextO = (SSS && bs.hasExit) ? sssExitOrigin(bs.exitPoint, n, bs.dir) : offsetOrigin(of, bs.dir);
                            nextD = bs.dir;
                            if (TEX && pool.cone) {
                                // the path's ray cone: width at this hit, spread widened by the sampled lobe (pathtrace.metal:7262-7267, 5703-5715)
                                if (!haveCone) cone = pool.cone[slot];
                                haveCone = true;
                                cone.x = smax(cone.x + cone.y * smax(hitv.x, 0.0f), 1.0e-7f);
                                float inc = 0.0f;
                                if (!bs.isDelta) {
                                    const bool pbr = cc.metalPbr && type == 7u;
                                    const int lobe = pbr ? bs.lobe : ((type == 0u || type == 5u) ? 0 : 1);
                                    const float r = clampf(pbr ? bs.lobeRoughness : mat.roughness01(), 0.0f, 1.0f);
                                    inc = lobe == 0 ? 0.55f : (lobe == 1 ? 0.03f + (0.45f - 0.03f) * r : 0.10f + (0.60f - 0.10f) * r);
                                }
                                cone.y = smin(cone.y + inc, 1.5f);
                            }
                            if (rp.enableRussianRoulette && depth >= 5u) {
                                const float p = clampf(maxComp, 0.05f, 0.95f);
                                if (rngNext(rng) > p) {
                                    endPath = true;
                                } else {
                                    thr /= p;
                                }
                            }
                        }
                    }
                    partEnd<COUNT>(counts, kShadePartBsdf, tBsdf);
                }
            }

            if (!(SSS && walking)) {
                ++depth;
                if (depth >= rp.maxDepth) endPath = true;
            }

            if (endPath) {
                // the item (one sample) is finished: ask for a new one (below, wave-wide)
                thr = mk3(1.0f);
                lastPdf = 1.0f;
                lastDelta = true;
                depth = 0u;
                specDepth = 0u;
                mediumDepth = 0u;
                needItem = true;
                if (want[0] | want[1] | want[2] | want[3] | want[4]) {
                    flushNext = true;       // publish this item after its last connections have been added
                    pool.flushItem[slot] = item;
                } else {
                    pool.itemAccum[item] = mk4(acc, __uint_as_float(sig));   // nothing outstanding: publish it now
                    acc = mk3(0.0f);
                    sig = 0u;
                }
            } else {
                stillAlive = true;
                walkFlag = SSS && walking;
            }
        }
    }

    if (COUNT) counts.stage[8] += needItem ? 1u : 0u;
    // ---- claim new work items ----
    const long long tItem = partBegin<COUNT>();
    if (MODE != kShadeDense) {
        // end of the frame: the range heads are dry; what can be left is the unused part of the last reservation of this slot's
        // 64-slot group (csrc/host/hip_backend.cpp hands a group to the tail kernel only after the heads ran dry)
        if (needItem) {
            uint2* const res = pool.itemReserve + slot / 64u;
            const uint32_t end = res->y;
            if (res->x < end) {
                const uint32_t got = atomicAdd(&res->x, 1u);
                if (got < end && got < rp.itemCount) {
                    item = got;
                    beginItem(rp, pool, got, rng, nextO, nextD);
                    stillAlive = true;
                    newSample = true;
                }
            }
        }
    } else {
        const uint32_t claimed = claimItems(rp, pool, slot, needItem, reservation);
        if (needItem && claimed < rp.itemCount) {
            item = claimed;
            beginItem(rp, pool, claimed, rng, nextO, nextD);
            stillAlive = true;
            newSample = true;
        }
    }

    partEnd<COUNT>(counts, kShadePartItem, tItem);
    const long long tStore = partBegin<COUNT>();
    // light connections of this bounce: k_connect walks the connect list k_shade appends to below
    uint32_t pendingMask = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kRecSlots; ++k) {
        if (want[k]) pendingMask |= 1u << k;
    }
    if (touched) {
        const uint32_t flags = (stillAlive ? kFlagAlive : 0u) | (lastDelta ? kFlagLastDelta : 0u) | (flushNext ? kFlagFlush : 0u) |
                               (walkFlag ? kFlagWalk : 0u) | (depth << kFlagDepthShift) | (specDepth << kFlagSpecDepthShift) |
                               (mediumDepth << kFlagMediumShift) | (pendingMask << kFlagPendingShift);
        pool.ray1[slot] = make_float4(nextD.y, nextD.z, lastPdf, __uint_as_float(flags));
        pool.accum[slot] = mk4(acc, __uint_as_float(item));
        if (stillAlive) {
            pool.ray0[slot] = mk4(nextO, nextD.x);
            pool.thr[slot] = mk4(thr, __uint_as_float(rng));
        }
        if (COUNT && pool.signature) pool.signature[slot] = sig;
        if (TEX && pool.cone && stillAlive && (haveCone || newSample)) pool.cone[slot] = newSample ? primaryCone(rp) : cone;
    }

    if (!TAIL && pool.connectList) {
        // one atomic per wave that queued anything, on the sub-list of this wave (see PathPool::connectList)
        const bool queued = touched && pendingMask != 0u;
        const unsigned long long mask = __ballot(queued);
        if (mask != 0ull) {
            const uint32_t queue = __builtin_amdgcn_readfirstlane(listWave) & (kConnectQueues - 1u);
            uint32_t base = 0u;
            if (laneId() == 0u) base = atomicAdd(pool.connectCount + queue * kConnectCountStride, static_cast<uint32_t>(__popcll(mask)));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << laneId()) - 1ull)));
            if (queued) pool.connectList[queue * pool.connectRegion + base + rank] = slot | (pendingMask << kConnectMaskShift);
        }
    }

    if (!TAIL && pool.busyOut) {
        // end of the frame: the slots the next iteration has to visit (same scheme)
        const bool busy = touched && (stillAlive || pendingMask != 0u || flushNext);
        const unsigned long long mask = __ballot(busy);
        if (mask != 0ull) {
            const uint32_t queue = __builtin_amdgcn_readfirstlane(listWave) & (kConnectQueues - 1u);
            uint32_t base = 0u;
            if (laneId() == 0u) base = atomicAdd(pool.busyCountOut + queue * kConnectCountStride, static_cast<uint32_t>(__popcll(mask)));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << laneId()) - 1ull)));
            if (busy) pool.busyOut[queue * pool.connectRegion + base + rank] = slot | (stillAlive ? kBusyAliveBit : 0u);
        }
    }

    partEnd<COUNT>(counts, kShadePartStore, tStore);
    if (COUNT) {
        counts.shadedHit += shadedHit;
        counts.triHit += triHit;
        counts.primary += primary;
    }
}

// LISTED: the launch walks pool.busyIn instead of the slots (end of the frame, see PathPool)
template <bool COUNT, bool SSS, bool TEX, bool LISTED, uint32_t MATS = kAllMaterials>
__global__ void __launch_bounds__(kShadeBlock) PTR_SHADE_WAVES_ATTR_M k_shade(RenderParams rp, SceneView sc, PathPool pool, ShadeResets resets) {
    const uint32_t index = blockIdx.x * kShadeBlock + threadIdx.x;
    if (index == 0u) {
        // k_shade runs between this iteration's k_extend and k_connect: it clears the work heads they will claim from
        // next (and the live-slot counter of the next k_extend), which saves two fill dispatches per iteration
        if (resets.extendHead) *resets.extendHead = 0u;
        if (resets.connectHead) *resets.connectHead = 0u;
        if (resets.nextAlive) *resets.nextAlive = 0u;
    }
    if (index < kConnectQueues) {   // the next iteration's list counters
        if (pool.connectClear) pool.connectClear[index * kConnectCountStride] = 0u;
        if (pool.busyCountClear) pool.busyCountClear[index * kConnectCountStride] = 0u;
    }
    ShadeCounts counts;
    if (LISTED) {
        SubLists lists;
        lists.init(pool.busyCountIn, pool.connectRegion);
        uint32_t slot = index;   // while the list is too long to beat a walk over the slots in order
        bool inRange = index < pool.slots, drained = resets.drained != 0u;
        if (static_cast<uint64_t>(lists.total) * PTR_BUSY_SHADE_DIV < pool.slots) {
            if ((index & ~63u) >= lists.total) return;   // the whole wave lies beyond the list
            inRange = index < lists.total;
            drained = false;
            const uint32_t at = lists.position(inRange ? index : 0u, pool.connectRegion);
            slot = inRange ? (pool.busyIn[at] & ~kBusyAliveBit) : 0u;
        }
        shadeSlot<COUNT, SSS, TEX, kShadeListed, MATS>(rp, sc, pool, slot, inRange, drained, index >> 6, counts);
    } else {
        shadeSlot<COUNT, SSS, TEX, kShadeDense, MATS>(rp, sc, pool, index, index < pool.slots, resets.drained != 0u, index >> 6, counts);
    }
    if (COUNT) {
        addCounter(pool.counters, kCntShadedHits, counts.shadedHit);
        addCounter(pool.counters, kCntTriangleHits, counts.triHit);
        addCounter(pool.counters, kCntPrimaryRays, counts.primary);
        addCounter(pool.counters, kCntExtendRays, counts.settled);
#pragma unroll
        for (uint32_t k = 0; k < 9u; ++k) addCounter(pool.counters, kCntShadeWaves + k, counts.stage[k]);
#pragma unroll
        for (uint32_t k = 0; k < kShadeParts; ++k) {
            addCounter(pool.counters, kCntShadeLaneTicks + k, counts.laneTicks[k]);
            addCounter(pool.counters, kCntShadeWaveTicks + k, counts.waveTicks[k]);
        }
    }
}

// =====================================================================================================
// k_connect: resolve queued light connections
// =====================================================================================================
namespace {

template <bool COUNT>
__device__ __forceinline__ f3 alongRect(const RenderParams& rp, const SceneView& sc, const ClampCfg& cc, f3 org, f3 dir, f3 weight,
                                        float bsdfPdfIn, f3 thr, LaneStack& stack, TraceCounters& cnt) {
    const TraceHit h = traverse<false, COUNT>(sc, org, dir, kEps, INFINITY, stack, cnt);
    if (h.prim == kHitMiss) return mk3(0.0f);
    const Surface ls = reconstruct(sc, org, dir, h.t, h.prim);
    f3 emission;
    float pdf;
    if (!rectLightHit(sc, ls, org, rp.emissionScale, emission, pdf)) return mk3(0.0f);
    const float lightPdf = smax(pdf, kSpecNeePdfFloor);
    const float invLightPdf = smin(1.0f / lightPdf, kSpecNeeInvPdfClamp);
    const float bsdfPdf = smax(bsdfPdfIn, kSpecNeePdfFloor);
    const float denom = lightPdf + bsdfPdf;
    float mis = denom > 0.0f ? (lightPdf / denom) : 0.0f;
    mis = clampf(mis, kMisMin, kMisMax);
    const f3 contrib = (weight * emission) * (mis * invLightPdf);
    return finite3(contrib) ? clampFirefly(thr, contrib, cc) : mk3(0.0f);
}

template <bool COUNT>
__device__ __forceinline__ f3 alongEnv(const RenderParams& rp, const SceneView& sc, const ClampCfg& cc, f3 org, f3 dir, f3 weight,
                                       float bsdfPdfIn, f3 thr, LaneStack& stack, TraceCounters& cnt) {
    const TraceHit h = traverse<true, COUNT>(sc, org, dir, kEps, INFINITY, stack, cnt);
    if (h.prim != kHitMiss) return mk3(0.0f);
    const float envPdf = smax(envPdfOf(sc, dir, rp.envRotation), kSpecNeePdfFloor);
    const float invEnvPdf = smin(1.0f / envPdf, kSpecNeeInvPdfClamp);
    const float bsdfPdf = smax(bsdfPdfIn, kSpecNeePdfFloor);
    const float denom = envPdf + bsdfPdf;
    float mis = denom > 0.0f ? (envPdf / denom) : 0.0f;
    mis = clampf(mis, kMisMin, kMisMax);
    const f3 envColor = envLookup(sc, dir, rp.envRotation, rp.envIntensity);
    const f3 contrib = (weight * envColor) * (mis * invEnvPdf);
    return finite3(contrib) ? clampFirefly(thr, contrib, cc) : mk3(0.0f);
}



// Contribution of a specular-NEE ray whose closest hit `h` has been found (kind 1 records).
__device__ __forceinline__ f3 rectContribution(const RenderParams& rp, const SceneView& sc, const ClampCfg& cc, f3 org, f3 dir,
                                               const TraceHit& h, f3 weight, float bsdfPdfIn, f3 thr) {
    if (h.prim == kHitMiss) return mk3(0.0f);
    const Surface ls = reconstruct(sc, org, dir, h.t, h.prim);
    return rectContributionAt(rp, sc, cc, ls, org, weight, bsdfPdfIn, thr);
}

// MNEE second bounce (kind 2 records): follow the specular ray to the next delta surface, scatter with a copy
// of the rng, then look for the environment / a rectangle light along the second specular direction.
template <bool COUNT>
__device__ f3 mneeChain(const RenderParams& rp, const SceneView& sc, const ClampCfg& cc, f3 org, f3 dir, uint32_t rngCopy, f3 weight,
                        float bsdfPdf, f3 thr, LaneStack& stack, TraceCounters& cntAny, TraceCounters& cntClosest, uint32_t& raysAny,
                        uint32_t& raysClosest) {
    f3 result = mk3(0.0f);
    if (COUNT) ++raysClosest;
    const TraceHit h = traverse<false, COUNT>(sc, org, dir, kEps, INFINITY, stack, cntClosest);
    if (h.prim == kHitMiss || sc.materialCount == 0u) return result;
    const Surface cs = reconstruct(sc, org, dir, h.t, h.prim);
    f3 tmpE;
    float tmpP;
    if (sc.rectLightCount > 0u && rectLightHit(sc, cs, org, rp.emissionScale, tmpE, tmpP)) return result;
    const Mat cm{sc.materials + static_cast<size_t>(min(cs.material, sc.materialCount - 1u)) * kMaterialVec4};
    if (!materialIsDelta(cm)) return result;
    f3 cn = cs.normal;
    if (dot(cn, cn) <= 0.0f) cn = mk3(0.0f, 1.0f, 0.0f);
    cn = normalize(cn);
    const f3 cin = normalize(dir);
    const BsdfSampleResult s2 = sampleBsdf(cm, cs.position, cn, -cin, cin, cs.frontFace, rngCopy, cc);
    if (!(s2.pdf > 0.0f && s2.isDelta && dot(s2.dir, s2.dir) > 0.0f && finite3(s2.weight))) return result;
    const f3 d2 = normalize(s2.dir);
    const f3 o2 = offsetOrigin(cs, d2);
    const f3 w2 = weight * s2.weight;
    const float pdf2 = bsdfPdf * s2.pdf;
    if (sc.envSampling) {
        if (COUNT) ++raysAny;
        result += alongEnv<COUNT>(rp, sc, cc, o2, d2, w2, pdf2, thr, stack, cntAny);
    }
    if (sc.rectLightCount > 0u) {
        if (COUNT) ++raysClosest;
        result += alongRect<COUNT>(rp, sc, cc, o2, d2, w2, pdf2, thr, stack, cntClosest);
    }
    return result;
}

}  // namespace

template <bool COUNT, int NODES>
__global__ void __launch_bounds__(kTraceBlock) PTR_EXTEND_ATTR k_connect(RenderParams rp, SceneView sc, PathPool pool, uint32_t* spill, uint32_t spillStride,
                                                          uint32_t* workCounter, int kRefillBelow, uint32_t feederChunk) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    stack.sp = 0u;
    TraceCounters cnt{0u, 0u}, cntClosest{0u, 0u};
    uint32_t rays = 0u, raysClosest = 0u, early = 0u;
    const ClampCfg cc = clampCfg(rp);

    // The work is the connect list k_shade filled (PathPool::connectList): lane l reads the counter of sub-list l, a wave prefix sum
    // gives every sub-list its place in one dense index space, and a lane finds the sub-list of an index with six cross-lane reads.
    // (Round 1 probed every slot's pending byte instead: 3.5 slots per ray found on config 2, a third of the kernel's instructions.)
    SceneMem mem = sceneMem(sc);
    SubLists lists;
    lists.init(pool.connectCount, pool.connectRegion);
    WaveFeeder feeder;
    // a list with fewer than 256 entries per resident wave is dealt out in smaller chunks, down to one wave-load each: on a scene that
    // queues few connections (config 5: 0.2 M records per launch over a 29 M-triangle tree) chunks of 256 put four latency-bound
    // batches one after the other on a quarter of the waves while the others had nothing
#ifdef PTR_CONNECT_CHUNK_FIXED   // (A/B switch)
    const uint32_t connectChunk = 256u;
#else
    const uint32_t connectWaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t connectChunk = min(256u, max(64u, ((lists.total + connectWaves - 1u) / connectWaves + 63u) & ~63u));
#endif
    feeder.init(workCounter, lists.total, connectChunk);
    Trav t;
    t.cur = 0u;
    bool active = false;
    uint32_t mySlot = 0u, bits = 0u;
    // the record arrays are one allocation: field f of record slot k lives at recBase[(k*4 + f)*slots + slot]
    float4* const recBase = pool.rec[0].org;
    const uint32_t slots = pool.recStride;   // field stride (whole pool), not the slot count of this group
    uint32_t myRecAt = 0u;   // (record*4)*slots + slot of the record this lane is resolving
    while (true) {
        const int nActive = __popcll(__ballot(active));
        const bool idleBits = __ballot(!active && bits != 0u) != 0ull;
        if (nActive < kRefillBelow && (idleBits || !feeder.exhausted)) {
            // batched refill (like k_extend): new slots for lanes that have nothing left, then the next record of
            // every idle lane.  Starting records lane by lane as they finish stalled the whole wave on each load.
            if (!feeder.exhausted) {
                const uint32_t idx = feeder.take(!active && bits == 0u);
                const uint32_t at = lists.position(idx != WaveFeeder::kNone ? idx : 0u, pool.connectRegion);
                if (idx != WaveFeeder::kNone) {
                    const uint32_t entry = pool.connectList[at];
                    mySlot = entry & ((1u << kConnectMaskShift) - 1u);
                    bits = entry >> kConnectMaskShift;
                }
            }
            if (!active && bits != 0u) {
                const uint32_t rec = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;
                bits &= bits - 1u;
                myRecAt = rec * 4u * slots + mySlot;
                const float4 o4 = recBase[myRecAt], d4 = recBase[myRecAt + slots];
                const uint32_t kind = __float_as_uint(d4.w);
                if (kind != 2u) {   // kind 2 (MNEE chains) is resolved by k_connect_chain
                    const bool any = kind != 1u;   // kind 0 and kind 3 (any-hit up to o4.w; kind 3 ignores one rectangle's triangles)
                    if (COUNT) { if (any) ++rays; else ++raysClosest; }
                    active = travBegin<NODES>(sc, t, mk3(o4), mk3(d4), kEps, any ? o4.w : INFINITY, any, stack);
                    if (kind == 3u) t.hit.prim = __float_as_uint(recBase[myRecAt + 3u * slots].x);
                    if (!active && kind == 1u) recBase[myRecAt + 2u * slots] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            continue;
        }
        if (nActive == 0) break;
        {
            // counting build: nodes/prims of closest-hit (kind 1) rays are booked with the extend counters
            TraceCounters step{0u, 0u};
            const bool more = travVote<COUNT, NODES>(sc, mem, t, active, stack, step);
            if (COUNT) {
                TraceCounters& dst = t.anyHit ? cnt : cntClosest;
                dst.nodes += step.nodes;
                dst.prims += step.prims;
            }
            if (!more) {
                active = false;
                float4* const a = recBase + myRecAt + 2u * slots;
                if (t.anyHit) {
                    if (COUNT) early += anyHitFound(t.hit.prim) ? 1u : 0u;
                    if (anyHitFound(t.hit.prim)) *a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else {
                    const float4 a4 = *a;
                    const f3 c = rectContribution(rp, sc, cc, t.org, t.dir, t.hit, mk3(a4), a4.w, mk3(recBase[myRecAt + 3u * slots]));
                    *a = mk4(c, 0.0f);
                }
            }
        }
    }
    if (COUNT) {
        addCounter(pool.counters, kCntShadowRays, rays);
        addCounter(pool.counters, kCntShadowNodes, cnt.nodes);
        addCounter(pool.counters, kCntShadowPrims, cnt.prims);
        addCounter(pool.counters, kCntShadowEarlyExit, early);
        addCounter(pool.counters, kCntExtendRays, raysClosest);
        addCounter(pool.counters, kCntExtendNodes, cntClosest.nodes);
        addCounter(pool.counters, kCntExtendPrims, cntClosest.prims);
    }
}

// MNEE two-bounce chains (record slot 4, kind 2; only launched when enableMnee && enableMneeSecondary).
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock) k_connect_chain(RenderParams rp, SceneView sc, PathPool pool, uint32_t* spill, uint32_t spillStride) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    stack.sp = 0u;
    TraceCounters cnt{0u, 0u}, cntClosest{0u, 0u};
    uint32_t rays = 0u, raysClosest = 0u;
    const ClampCfg cc = clampCfg(rp);
    for (uint32_t slot = gtid; slot < pool.slots; slot += gridDim.x * kTraceBlock) {
        if (!((__float_as_uint(pool.ray1[slot].w) >> kFlagPendingShift) & (1u << 4))) continue;
        const ShadowRecordView& r = pool.rec[4];
        const float4 d4 = r.dir[slot];
        if (__float_as_uint(d4.w) != 2u) continue;
        const float4 o4 = r.org[slot], a4 = r.a[slot];
        const f3 c = mneeChain<COUNT>(rp, sc, cc, mk3(o4), mk3(d4), __float_as_uint(o4.w), mk3(a4), a4.w, mk3(r.b[slot]), stack, cnt,
                                      cntClosest, rays, raysClosest);
        r.a[slot] = mk4(c, 0.0f);
    }
    if (COUNT) {
        addCounter(pool.counters, kCntShadowRays, rays);
        addCounter(pool.counters, kCntShadowNodes, cnt.nodes);
        addCounter(pool.counters, kCntShadowPrims, cnt.prims);
        addCounter(pool.counters, kCntExtendRays, raysClosest);
        addCounter(pool.counters, kCntExtendNodes, cntClosest.nodes);
        addCounter(pool.counters, kCntExtendPrims, cntClosest.prims);
    }
}

// =====================================================================================================
// End of the frame.  Once the item queue is dry every slot still finishes its path, at falling occupancy: a dozen
// extend / shade / connect rounds over a pool that is mostly dead, each paying its launches, its scan of dead slots and the
// latency of its longest ray (about 10 ms per frame in round 1, whatever the frame size).  When few slots are left the host
// hands them to these two kernels instead: k_tail_collect compacts the indices of the busy slots into a list (one atomic per
// wave), and k_tail_run gives every lane ONE slot and runs it to the end of its path - closest hit, shade, light
// connections, next bounce - through the same device functions as the three wavefront kernels, with no launch in between.
// Lanes diverge completely; with a few hundred thousand paths left that costs less than the launches did.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_tail_collect(PathPool pool, uint32_t* list, uint32_t* listCount) {
    __shared__ uint32_t found[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = laneId();
    const uint32_t wavesTotal = gridDim.x * 4u;
    const uint32_t waveId = blockIdx.x * 4u + wave;
    // each wave scans a contiguous range of slots, 64 at a time, and appends the busy ones to the list in batches of up to 64
    const uint32_t perWave = ((pool.slots + wavesTotal - 1u) / wavesTotal + 63u) & ~63u;
    const uint32_t begin = waveId * perWave, end = min(begin + perWave, pool.slots);
    uint32_t have = 0u;
    auto flush = [&]() {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(listCount, have);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < have) list[base + lane] = found[wave][lane];
        have = 0u;
    };
    for (uint32_t first = begin; first < end; first += 64u) {
        const uint32_t slot = first + lane;
        bool busy = false;
        if (slot < end) {
            const uint32_t flags = __float_as_uint(pool.ray1[slot].w);
            busy = (flags & (kFlagAlive | kFlagFlush | (kFlagPendingMask << kFlagPendingShift))) != 0u;
        }
        const unsigned long long mask = __ballot(busy);
        const uint32_t n = static_cast<uint32_t>(__popcll(mask));
        if (n == 0u) continue;
        if (have + n > 64u) flush();
        const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
        if (busy) found[wave][have + rank] = slot;
        have += n;
    }
    if (have) flush();
}

template <bool COUNT, bool SSS, bool TEX>
__global__ void __launch_bounds__(kTraceBlock) k_tail_run(RenderParams rp, SceneView sc, PathPool pool, const uint32_t* list, const uint32_t* listCount,
                                                          uint32_t* listHead, uint32_t* spill, uint32_t spillStride) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    stack.sp = 0u;
    const ClampCfg cc = clampCfg<SSS>(rp);
    TraceCounters cntExtend{0u, 0u}, cntAny{0u, 0u}, cntClosest{0u, 0u};
    uint32_t raysExtend = 0u, raysAny = 0u, raysClosest = 0u, early = 0u;
    ShadeCounts counts;
    const uint32_t total = *listCount;
    float4* const recBase = pool.rec[0].org;
    const uint32_t slots = pool.recStride;
    // a path never needs more visits than this (depth limit, plus the visit that publishes the item; random walks add their steps)
    const uint32_t visitLimit = (rp.maxDepth + 2u) * (1u + (SSS ? rp.sssMaxSteps : 0u)) + 2u;
    while (true) {
        uint32_t base = 0u;
        if (laneId() == 0u) base = atomicAdd(listHead, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= total) break;
        const uint32_t index = base + laneId();
        if (index < total) {
            const uint32_t slot = list[index];
            for (uint32_t visit = 0u; visit < visitLimit; ++visit) {
                const float4 r1 = pool.ray1[slot];
                const uint32_t flags = __float_as_uint(r1.w);
                if ((flags & (kFlagAlive | kFlagFlush | (kFlagPendingMask << kFlagPendingShift))) == 0u) break;
                if (flags & kFlagAlive) {
                    // k_extend's part
                    const float4 r0 = pool.ray0[slot];
                    if (COUNT) ++raysExtend;
                    const TraceHit h = traverse<false, COUNT>(sc, mk3(r0), mk3(r0.w, r1.x, r1.y), kEps, INFINITY, stack, cntExtend);
                    pool.hit[slot] = make_float2(h.t, __uint_as_float(h.prim));
                    __threadfence();
                }
                // k_shade's part
                shadeSlot<COUNT, SSS, TEX, kShadeTail>(rp, sc, pool, slot, true, false, 0u, counts);
                __threadfence();
                // k_connect's part: the records this visit queued
                uint32_t bits = (__float_as_uint(pool.ray1[slot].w) >> kFlagPendingShift) & kFlagPendingMask;
                while (bits != 0u) {
                    const uint32_t rec = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;
                    bits &= bits - 1u;
                    const uint32_t recAt = rec * 4u * slots + slot;
                    const float4 o4 = recBase[recAt], d4 = recBase[recAt + slots];
                    float4* const a = recBase + recAt + 2u * slots;
                    const uint32_t kind = __float_as_uint(d4.w);
                    if (kind == 0u || kind == 3u) {
                        if (COUNT) ++raysAny;
                        const uint32_t ignore = kind == 3u ? __float_as_uint(recBase[recAt + 3u * slots].x) : kHitMiss;
                        const TraceHit h = traverse<true, COUNT>(sc, mk3(o4), mk3(d4), kEps, o4.w, stack, cntAny, ignore);
                        if (COUNT) early += anyHitFound(h.prim) ? 1u : 0u;
                        if (anyHitFound(h.prim)) *a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else if (kind == 1u) {
                        if (COUNT) ++raysClosest;
                        const TraceHit h = traverse<false, COUNT>(sc, mk3(o4), mk3(d4), kEps, INFINITY, stack, cntClosest);
                        const float4 a4 = *a;
                        const f3 c = rectContribution(rp, sc, cc, mk3(o4), mk3(d4), h, mk3(a4), a4.w, mk3(recBase[recAt + 3u * slots]));
                        *a = mk4(c, 0.0f);
                    } else {
                        const float4 a4 = *a;
                        const f3 c = mneeChain<COUNT>(rp, sc, cc, mk3(o4), mk3(d4), __float_as_uint(o4.w), mk3(a4), a4.w, mk3(recBase[recAt + 3u * slots]),
                                                      stack, cntAny, cntClosest, raysAny, raysClosest);
                        *a = mk4(c, 0.0f);
                    }
                }
                __threadfence();
            }
        }
    }
    if (COUNT) {
        addCounter(pool.counters, kCntExtendRays, raysExtend + raysClosest);
        addCounter(pool.counters, kCntExtendNodes, cntExtend.nodes + cntClosest.nodes);
        addCounter(pool.counters, kCntExtendPrims, cntExtend.prims + cntClosest.prims);
        addCounter(pool.counters, kCntShadowRays, raysAny);
        addCounter(pool.counters, kCntShadowNodes, cntAny.nodes);
        addCounter(pool.counters, kCntShadowPrims, cntAny.prims);
        addCounter(pool.counters, kCntShadowEarlyExit, early);
        addCounter(pool.counters, kCntShadedHits, counts.shadedHit);
        addCounter(pool.counters, kCntTriangleHits, counts.triHit);
        addCounter(pool.counters, kCntPrimaryRays, counts.primary);
        addCounter(pool.counters, kCntExtendRays, counts.settled);
    }
}

// =====================================================================================================
// k_resolve: fixed-order per-pixel reduction
// =====================================================================================================
__global__ void __launch_bounds__(256) k_flush(RenderParams rp, PathPool pool) {
    // after the last bounce: add the connections still outstanding and publish the items that were waiting for them
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= pool.slots) return;
    const uint32_t flags = __float_as_uint(pool.ray1[slot].w);
    if (!(flags & kFlagFlush)) return;
    const uint32_t pendingIn = (flags >> kFlagPendingShift) & kFlagPendingMask;
    f3 acc = mk3(pool.accum[slot]);
    uint32_t sig = pool.signature ? pool.signature[slot] : 0u;
    for (uint32_t k = 0; k < kRecSlots; ++k) {
        if (!(pendingIn & (1u << k))) continue;
        const float4 a = pool.rec[k].a[slot];
        acc += mk3(a);
        if (k == 0u && (a.x > 0.0f || a.y > 0.0f || a.z > 0.0f) && static_cast<uint32_t>(a.w) < kSigNeeBits) sig |= 1u << static_cast<uint32_t>(a.w);
    }
    pool.itemAccum[pool.flushItem[slot]] = mk4(acc, __uint_as_float(sig));
    (void)rp;
}

__global__ void __launch_bounds__(256) k_resolve(RenderParams rp, PathPool pool, uint32_t partCount, float* out) {
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= rp.localPixels) return;
    f3 sum = mk3(0.0f);
    for (uint32_t c = 0; c < rp.spp; ++c) sum += mk3(pool.itemAccum[static_cast<size_t>(c) * rp.localPixels + lp]);
    const uint32_t pixel = pool.pixelOfLocal[lp];
    const uint32_t x = pixel % rp.width, y = pixel / rp.width;
    const uint32_t localBand = (y / PTR_BAND_ROWS) / partCount;
    float* o = out + (static_cast<size_t>(localBand * PTR_BAND_ROWS + (y % PTR_BAND_ROWS)) * rp.width + x) * 3u;
    // a frame rendered in several passes keeps its running sum in the output buffer between them
    f3 total = (rp.passFlags & 1u) ? sum : mk3(o[0], o[1], o[2]) + sum;
    if (rp.passFlags & 2u) total = total / static_cast<float>(rp.sppTotal);
    o[0] = total.x;
    o[1] = total.y;
    o[2] = total.z;
}

// =====================================================================================================
// k_interleave_bands: the gather step of a multi-device frame.  `gathered` holds the band buffers of the P partitions one
// after the other (partition p starts at float offset partOffset[p]); image band b = local band b / P of partition b % P.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_interleave_bands(const float* gathered, const uint64_t* partOffset, uint32_t parts, uint32_t width,
                                                          uint32_t height, float* image) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // one float of the image
    const uint64_t rowFloats = static_cast<uint64_t>(width) * 3u;
    if (i >= rowFloats * height) return;
    const uint32_t y = static_cast<uint32_t>(i / rowFloats);
    const uint64_t inRow = i - static_cast<uint64_t>(y) * rowFloats;
    const uint32_t band = y / PTR_BAND_ROWS, part = band % parts, localBand = band / parts;
    image[i] = gathered[partOffset[part] + (static_cast<uint64_t>(localBand) * PTR_BAND_ROWS + (y % PTR_BAND_ROWS)) * rowFloats + inRow];
}

// =====================================================================================================
// k_aovs: first-hit feature buffers (what the reference hands to its denoiser: shaders/pathtrace.metal:6424-6435 sets
// albedo = material base colour and normal = shading normal at the first hit; 9813-9815 writes albedo and
// normal*0.5+0.5 per pixel).  One camera ray per pixel, the one sample `sample` of the path tracer starts with.
// albedo.w = 1 on a hit, 0 on a miss; normal.w = hit distance (0 on a miss).
// =====================================================================================================
__global__ void __launch_bounds__(kTraceBlock) k_aovs(RenderParams rp, SceneView sc, uint32_t sample, float4* albedo, float4* normal, uint32_t* spill,
                                                       uint32_t spillStride) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    TraceCounters cnt{0u, 0u};
    const uint32_t pixels = rp.width * rp.height;
    for (uint32_t pixel = gtid; pixel < pixels; pixel += gridDim.x * kTraceBlock) {
        uint32_t rng;
        f3 org, dir;
        beginSample(rp, pixel, sample, rng, org, dir);
        const TraceHit h = traverse<false, false>(sc, org, dir, kEps, INFINITY, stack, cnt);
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n = make_float4(0.5f, 0.5f, 0.5f, 0.0f);
        if (h.prim != kHitMiss && sc.materialCount > 0u) {
            const Surface sf = reconstruct(sc, org, dir, h.t, h.prim);
            const Mat mat{sc.materials + static_cast<size_t>(min(sf.material, sc.materialCount - 1u)) * kMaterialVec4};
            f3 sn = sf.hitShadingNormal;
            if (dot(sn, sn) <= 0.0f) sn = sf.normal;
            if (mat.type() == 2u) sn = sf.normal;
            sn = normalize(sn);
            a = mk4(mat.baseColor(), 1.0f);
            n = mk4(sn * 0.5f + mk3(0.5f), h.t);
        }
        albedo[pixel] = a;
        normal[pixel] = n;
    }
}

// =====================================================================================================
// k_trace_rays: ray-batch queries for ptr_trace_rays (parity tests against the oracle ray caster)
// =====================================================================================================
template <bool ANY>
__global__ void __launch_bounds__(kTraceBlock) k_trace_rays(SceneView sc, const float4* rays, uint64_t n, PtrHit* out, uint32_t* spill,
                                                             uint32_t spillStride, uint64_t* counters) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    TraceCounters cnt{0u, 0u};
    for (uint64_t i = gtid; i < n; i += static_cast<uint64_t>(gridDim.x) * kTraceBlock) {
        const float4 a = rays[i * 2u], b = rays[i * 2u + 1u];
        const f3 org = mk3(a), dir = mk3(b);
        const TraceHit h = traverse<ANY, true>(sc, org, dir, a.w, b.w, stack, cnt);
        float hu = 0.0f, hv = 0.0f;
        PtrHit r;
        r.t = -1.0f;
        r.u = 0.0f;
        r.v = 0.0f;
        r.primType = 0u;
        r.geomIndex = 0u;
        r.primIndex = 0u;
        r.ng[0] = r.ng[1] = r.ng[2] = 0.0f;
        r.pad = 0u;
        if (h.prim != kHitMiss) {
            r.t = ANY ? 0.0f : h.t;
            if (!ANY) {
                if (h.prim & kHitSphereBit) {
                    r.primType = 1u;
                    r.primIndex = sc.sphereInfo[h.prim & kHitIndexMask].x;
                } else {
                    const float4* tp = sc.tris + static_cast<size_t>(h.prim & kHitIndexMask) * 3u;
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    const uint32_t meta = __float_as_uint(t1.w);
                    triangleUv(mk3(t0), mk3(t1), mk3(t2), org, dir, hu, hv);
                    r.u = hu;
                    r.v = hv;
                    const f3 ng = cross(mk3(t2), mk3(t1));
                    r.ng[0] = ng.x;
                    r.ng[1] = ng.y;
                    r.ng[2] = ng.z;
                    if ((meta >> 30) == 0u) {
                        r.primType = 0u;
                        r.geomIndex = meta & kTriGeomMask;
                        r.primIndex = __float_as_uint(t2.w);
                    } else {
                        r.primType = 2u;
                        r.primIndex = meta & kTriGeomMask;
                    }
                }
            }
        }
        out[i] = r;
    }
    if (counters) {
        addCounter(counters, ANY ? kCntShadowNodes : kCntExtendNodes, cnt.nodes);
        addCounter(counters, ANY ? kCntShadowPrims : kCntExtendPrims, cnt.prims);
    }
}

// =====================================================================================================
// debug kernels (known-answer tests of device functions)
// =====================================================================================================
__global__ void k_debug_eval(const float4* material, RenderParams rp, const float* in, uint64_t n, float* out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + i * 12u;
    const Mat m{material};
    const BsdfEvalResult e = evalBsdf<true>(m, ld3(p), ld3(p + 3), ld3(p + 6), ld3(p + 9), clampCfg(rp));   // (the run-time flag still decides)
    float* o = out + i * 5u;
    o[0] = e.value.x; o[1] = e.value.y; o[2] = e.value.z; o[3] = e.pdf; o[4] = e.isDelta ? 1.0f : 0.0f;
}

__global__ void k_debug_sample(const float4* material, RenderParams rp, const float* in, const uint32_t* front, const uint32_t* rngIn,
                               uint64_t n, float* out, uint32_t* rngOut) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + i * 9u;
    const Mat m{material};
    uint32_t rng = rngIn[i];
    const f3 wo = ld3(p + 6);
    const BsdfSampleResult s = sampleBsdf<true>(m, ld3(p), ld3(p + 3), wo, -wo, front[i] != 0u, rng, clampCfg(rp));
    float* o = out + i * 8u;
    o[0] = s.dir.x; o[1] = s.dir.y; o[2] = s.dir.z;
    o[3] = s.weight.x; o[4] = s.weight.y; o[5] = s.weight.z;
    o[6] = s.pdf; o[7] = s.isDelta ? 1.0f : 0.0f;
    rngOut[i] = rng;
}

__global__ void k_debug_tex_sample(SceneView sc, uint32_t texture, const float* in, uint64_t n, float4* out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = texSample(sc, texture, in[i * 3u], in[i * 3u + 1u], in[i * 3u + 2u], make_float4(-1.0f, -1.0f, -1.0f, -1.0f));
}

// closest hit + surface reconstruction + next-ray origin, the pieces k_shade builds a bounce from (tests of a13 / a14)
__global__ void __launch_bounds__(kTraceBlock) k_debug_surface(SceneView sc, const float* in, uint64_t n, float* out, uint32_t* spill, uint32_t spillStride) {
    __shared__ uint32_t ldsStack[kLdsStackLevels * kTraceBlock];
    const uint32_t gtid = blockIdx.x * kTraceBlock + threadIdx.x;
    LaneStack stack;
    stack.lds = (LdsWord*)(ldsStack + threadIdx.x);
    stack.spill = spill;
    stack.spillStride = spillStride;
    stack.limit = sc.stackLimit;
    TraceCounters cnt{0u, 0u};
    for (uint64_t i = gtid; i < n; i += static_cast<uint64_t>(gridDim.x) * kTraceBlock) {
        const float* r = in + i * 9u;
        float* o = out + i * 16u;
        for (int k = 0; k < 16; ++k) o[k] = 0.0f;
        const f3 org = ld3(r), dir = ld3(r + 3);
        const TraceHit h = traverse<false, false>(sc, org, dir, kEps, INFINITY, stack, cnt);
        if (h.prim == kHitMiss) continue;
        const Surface sf = reconstruct(sc, org, dir, h.t, h.prim);
        const f3 next = offsetOrigin(sf, ld3(r + 6));
        o[0] = 1.0f;
        o[1] = sf.t;
        o[2] = sf.position.x, o[3] = sf.position.y, o[4] = sf.position.z;
        o[5] = sf.normal.x, o[6] = sf.normal.y, o[7] = sf.normal.z;
        o[8] = sf.hitShadingNormal.x, o[9] = sf.hitShadingNormal.y, o[10] = sf.hitShadingNormal.z;
        o[11] = sf.frontFace ? 1.0f : 0.0f;
        o[12] = next.x, o[13] = next.y, o[14] = next.z;
    }
}

__global__ void k_debug_camera(RenderParams rp, const uint32_t* xys, uint64_t n, float* out, uint32_t* rngOut) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = xys[i * 3u], y = xys[i * 3u + 1u], s = xys[i * 3u + 2u];
    uint32_t rng;
    f3 o, d;
    beginSample(rp, y * rp.width + x, s, rng, o, d);
    float* q = out + i * 6u;
    q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = d.x; q[4] = d.y; q[5] = d.z;
    rngOut[i] = rng;
}

// =====================================================================================================
// launchers
// =====================================================================================================
static inline uint32_t ceilDiv(uint64_t a, uint32_t b) { return static_cast<uint32_t>((a + b - 1) / b); }

// The host sizes the traversal grids in units of kTraceGridUnit threads; the kernels may be built with smaller blocks (PTR_TRACE_BLOCK in
// bvh_layout.h): same number of resident threads, so the spill area and its stride are unchanged.
static inline LaunchConfig perBlockSize(LaunchConfig cfg) {
    cfg.traceGrid *= kTraceGridUnit / kTraceBlock;
    return cfg;
}

void launchGenerate(const RenderParams& rp, const PathPool& pool, hipStream_t stream) {
    hipLaunchKernelGGL(k_generate, dim3(ceilDiv(pool.slots, 256)), dim3(256), 0, stream, rp, pool);
}

void launchExtend(const SceneView& sc, const PathPool& pool, const LaunchConfig& cfgIn, uint32_t* aliveOut, bool count, hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    const uint32_t grid = std::min(cfg.traceGrid, ceilDiv(pool.slots, kTraceBlock));
    // the live-slot count is a separate instantiation so the common launch carries no extra register
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kTraceBlock), 0, stream, sc, pool, cfg.spill, stride, cfg.workCounters, cfg.refillBelow,
                           cfg.feederChunk, aliveOut);
    };
    // the node format is a compile-time choice of the persistent kernels (no dead float / quantised path in the step loop); the
    // counting build keeps the run-time flag
    if (count) {
        if (aliveOut) launch(k_extend<true, true, -1>); else launch(k_extend<true, false, -1>);
    } else if (sc.useQuantized && sc.useWide) {
        if (aliveOut) launch(k_extend<false, true, 2>); else launch(k_extend<false, false, 2>);
    } else if (sc.useQuantized) {
        if (aliveOut) launch(k_extend<false, true, 1>); else launch(k_extend<false, false, 1>);
    } else {
        if (aliveOut) launch(k_extend<false, true, 0>); else launch(k_extend<false, false, 0>);
    }
}


// The k_shade instantiation a render of this scene launches: the smallest compiled set (materials + features) that covers it, or
// kAllMaterials.  The Metal-semantics instantiations (SSS, TEX) and the counting ones are compiled for every material.
uint32_t shadeKernelSet(const RenderParams& rp, const SceneView& sc, bool count) {
    const bool sss = (rp.mediaMode & (PTR_METAL_SSS | PTR_METAL_PBR | PTR_METAL_CLAMPS)) != 0u;
    if (sss || count || sc.materialTypes == 0u) return kAllMaterials;
    // (an environment map and the Metal media / face-normal rules are features of the set like a material type is)
    const uint32_t needs = sc.materialTypes | ((sc.envWidth > 0u || sc.envSampling) ? kFeatureEnvironment : 0u) |
                           ((rp.mediaMode & (PTR_METAL_MEDIA | PTR_METAL_FACE_NORMAL)) ? kFeatureMedia : 0u);
    for (uint32_t set : {kDiffuseMaterials, kBasicMaterials, kMetalMaterials, kCarPaintMaterials, kPbrMaterials}) {
        if ((needs & ~set) == 0u) return set;
    }
    return kAllMaterials;
}

void launchShade(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const ShadeResets& resets, bool count,
                 hipStream_t stream) {
    const bool sss = (rp.mediaMode & (PTR_METAL_SSS | PTR_METAL_PBR | PTR_METAL_CLAMPS)) != 0u;   // the instantiation that carries those Metal-only models
    const bool listed = pool.busyIn != nullptr;   // the grid still covers every slot: waves beyond the list leave at once
    const bool tex = sss && sc.textureCount > 0u;   // the instantiation with the texture lookups and the ray cone
    const uint32_t grid = ceilDiv(pool.slots, kShadeBlock);
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(kShadeBlock), 0, stream, rp, sc, pool, resets); };
    const uint32_t set = shadeKernelSet(rp, sc, count);
    auto pick = [&](auto countTag, auto listedTag) {
        constexpr bool C = decltype(countTag)::value, L = decltype(listedTag)::value;
        if (tex) launch(k_shade<C, true, true, L>);
        else if (sss) launch(k_shade<C, true, false, L>);
        else if (C) launch(k_shade<C, false, false, L>);
        else if (set == kDiffuseMaterials) launch(k_shade<false, false, false, L, kDiffuseMaterials>);
        else if (set == kBasicMaterials) launch(k_shade<false, false, false, L, kBasicMaterials>);
        else if (set == kMetalMaterials) launch(k_shade<false, false, false, L, kMetalMaterials>);
        else if (set == kCarPaintMaterials) launch(k_shade<false, false, false, L, kCarPaintMaterials>);
        else if (set == kPbrMaterials) launch(k_shade<false, false, false, L, kPbrMaterials>);
        else launch(k_shade<C, false, false, L>);
    };
    if (count) {
        if (listed) pick(std::true_type{}, std::true_type{}); else pick(std::true_type{}, std::false_type{});
    } else {
        if (listed) pick(std::false_type{}, std::true_type{}); else pick(std::false_type{}, std::false_type{});
    }
}

void launchConnect(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const LaunchConfig& cfgIn, bool count, hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    {
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(cfg.traceGrid), dim3(kTraceBlock), 0, stream, rp, sc, pool, cfg.spill, stride, cfg.workCounters + 1,
                               cfg.refillBelow, cfg.feederChunk);
        };
        if (count) {
            launch(k_connect<true, -1>);
        } else if (sc.useQuantized && sc.useWide) {
            launch(k_connect<false, 2>);
        } else if (sc.useQuantized) {
            launch(k_connect<false, 1>);
        } else {
            launch(k_connect<false, 0>);
        }
    }
    if (rp.enableMnee && rp.enableMneeSecondary) {
        if (count) {
            hipLaunchKernelGGL(k_connect_chain<true>, dim3(cfg.traceGrid), dim3(kTraceBlock), 0, stream, rp, sc, pool, cfg.spill, stride);
        } else {
            hipLaunchKernelGGL(k_connect_chain<false>, dim3(cfg.traceGrid), dim3(kTraceBlock), 0, stream, rp, sc, pool, cfg.spill, stride);
        }
    }
}

void launchTail(const RenderParams& rp, const SceneView& sc, const PathPool& pool, const LaunchConfig& cfgIn, uint32_t* dList, uint32_t* dListCount,
                uint32_t* dListHead, bool count, hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    // dListCount and dListHead are zero on entry (the caller clears them on the same stream)
    const uint32_t collectGrid = std::max(1u, std::min(cfg.traceGrid, ceilDiv(pool.slots, 256u * 16u)));
    hipLaunchKernelGGL(k_tail_collect, dim3(collectGrid), dim3(256), 0, stream, pool, dList, dListCount);
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    const bool sss = (rp.mediaMode & (PTR_METAL_SSS | PTR_METAL_PBR | PTR_METAL_CLAMPS)) != 0u;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(cfg.traceGrid), dim3(kTraceBlock), 0, stream, rp, sc, pool, dList, dListCount, dListHead, cfg.spill, stride);
    };
    const bool tex = sss && sc.textureCount > 0u;
    if (count) {
        if (tex) launch(k_tail_run<true, true, true>); else if (sss) launch(k_tail_run<true, true, false>); else launch(k_tail_run<true, false, false>);
    } else {
        if (tex) launch(k_tail_run<false, true, true>); else if (sss) launch(k_tail_run<false, true, false>); else launch(k_tail_run<false, false, false>);
    }
}

void launchResolve(const RenderParams& rp, const PathPool& pool, uint32_t partCount, float* dOut, hipStream_t stream) {
    hipLaunchKernelGGL(k_flush, dim3(ceilDiv(pool.slots, 256)), dim3(256), 0, stream, rp, pool);
    hipLaunchKernelGGL(k_resolve, dim3(ceilDiv(rp.localPixels, 256)), dim3(256), 0, stream, rp, pool, partCount, dOut);
}

void launchInterleaveBands(const float* dGathered, const uint64_t* dPartOffset, uint32_t parts, uint32_t width, uint32_t height, float* dImage,
                           hipStream_t stream) {
    const uint64_t floats = static_cast<uint64_t>(width) * height * 3u;
    hipLaunchKernelGGL(k_interleave_bands, dim3(ceilDiv(floats, 256)), dim3(256), 0, stream, dGathered, dPartOffset, parts, width, height, dImage);
}

void launchTraceRays(const SceneView& sc, const float4* dRays, uint64_t n, bool anyHit, PtrHit* dOut, const LaunchConfig& cfgIn,
                     uint64_t* dCounters, hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    const uint32_t grid = std::min(cfg.traceGrid, std::max(1u, ceilDiv(n, kTraceBlock)));
    if (anyHit) {
        hipLaunchKernelGGL(k_trace_rays<true>, dim3(grid), dim3(kTraceBlock), 0, stream, sc, dRays, n, dOut, cfg.spill, stride, dCounters);
    } else {
        hipLaunchKernelGGL(k_trace_rays<false>, dim3(grid), dim3(kTraceBlock), 0, stream, sc, dRays, n, dOut, cfg.spill, stride, dCounters);
    }
}

void launchAovs(const RenderParams& rp, const SceneView& sc, uint32_t sample, float4* dAlbedo, float4* dNormal, const LaunchConfig& cfgIn,
                hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    const uint32_t pixels = rp.width * rp.height;
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    const uint32_t grid = std::min(cfg.traceGrid, std::max(1u, ceilDiv(pixels, kTraceBlock)));
    hipLaunchKernelGGL(k_aovs, dim3(grid), dim3(kTraceBlock), 0, stream, rp, sc, sample, dAlbedo, dNormal, cfg.spill, stride);
}

void launchDebugEvalBsdf(const float4* dMaterial, const RenderParams& rp, const float* dIn, uint64_t n, float* dOut, hipStream_t stream) {
    hipLaunchKernelGGL(k_debug_eval, dim3(std::max(1u, ceilDiv(n, 128))), dim3(128), 0, stream, dMaterial, rp, dIn, n, dOut);
}

void launchDebugSampleBsdf(const float4* dMaterial, const RenderParams& rp, const float* dIn, const uint32_t* dFront, const uint32_t* dRng,
                           uint64_t n, float* dOut, uint32_t* dRngOut, hipStream_t stream) {
    hipLaunchKernelGGL(k_debug_sample, dim3(std::max(1u, ceilDiv(n, 128))), dim3(128), 0, stream, dMaterial, rp, dIn, dFront, dRng, n, dOut, dRngOut);
}

void launchDebugTexSample(const SceneView& sc, uint32_t texture, const float* dIn, uint64_t n, float4* dOut, hipStream_t stream) {
    hipLaunchKernelGGL(k_debug_tex_sample, dim3(std::max(1u, ceilDiv(n, 128))), dim3(128), 0, stream, sc, texture, dIn, n, dOut);
}

void launchDebugSurfaceHits(const SceneView& sc, const float* dIn, uint64_t n, float* dOut, const LaunchConfig& cfgIn, hipStream_t stream) {
    const LaunchConfig cfg = perBlockSize(cfgIn);
    const uint32_t stride = cfg.traceGrid * kTraceBlock;
    const uint32_t grid = std::min(cfg.traceGrid, std::max(1u, ceilDiv(n, kTraceBlock)));
    hipLaunchKernelGGL(k_debug_surface, dim3(grid), dim3(kTraceBlock), 0, stream, sc, dIn, n, dOut, cfg.spill, stride);
}

void launchDebugCameraRays(const RenderParams& rp, const uint32_t* dXys, uint64_t n, float* dOut, uint32_t* dRngOut, hipStream_t stream) {
    hipLaunchKernelGGL(k_debug_camera, dim3(std::max(1u, ceilDiv(n, 128))), dim3(128), 0, stream, rp, dXys, n, dOut, dRngOut);
}

}  // namespace ptrk
