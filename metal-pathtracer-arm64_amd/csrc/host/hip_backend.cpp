// Device side of the C-ABI: scene upload (BVH build + SoA flattening + H2D), the wavefront render loop,
// and ray-batch queries.  Compiled with hipcc (host code only; kernels live in kernels/wavefront.hip).
//
// Reference counterparts: SceneResources::rebuildAccelerationStructures (src/renderer/SceneResources.mm:2055-2259),
// SoftwareBvhAccel::rebuild (src/renderer/SceneAccel.mm:23-325), the per-sample dispatch loop
// (src/renderer/RenderLoop.mm:366-386, src/headless/MetalHeadlessRenderer.mm:64-92) and the Embree backend's
// scene assembly (src/headless/EmbreeHeadlessRenderer.mm:2077-2300, 2484-2522).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../kernels/device_types.h"
#include "../kernels/launch.h"
#include "bvh_builder.h"
#include "env_importance_sampler.h"
#include "geometry_cache.h"
#include "knobs.h"
#include "parallel.h"
#include "ptr_abi.h"
#include "ptr_debug.h"
#include "scene_geometry.h"
#include "vecmath.h"

using namespace ptrk;

namespace {

struct HipError {
    std::string message;
};

#define HIP_CHECK(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            throw HipError{std::string(#expr) + ": " + hipGetErrorString(_e)};                              \
        }                                                                                                   \
    } while (0)

void setErr(char* err, size_t cap, const std::string& msg) {
    if (err && cap > 0) std::snprintf(err, cap, "%s", msg.c_str());
}

// Nothing may unwind across the C boundary: the BVH build allocates multi-GB vectors and starts threads
// (std::bad_alloc, std::system_error), and the callers are ctypes / a C++ program built with another runtime.
#define PTR_CATCH_ALL(err, cap)                                                                             \
    catch (const HipError& e) {                                                                             \
        setErr(err, cap, e.message);                                                                        \
        return 1;                                                                                           \
    }                                                                                                       \
    catch (const std::exception& e) {                                                                       \
        setErr(err, cap, std::string("exception: ") + e.what());                                            \
        return 1;                                                                                           \
    }                                                                                                       \
    catch (...) {                                                                                           \
        setErr(err, cap, "unknown exception");                                                              \
        return 1;                                                                                           \
    }

template <typename T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    void ensure(size_t n) {
        if (n <= count && ptr) return;
        release();
        if (n == 0) n = 1;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
        count = n;
    }
    void upload(const T* src, size_t n) {
        ensure(n);
        if (n) HIP_CHECK(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
};

using ptr::float3;

void put4(std::vector<float>& dst, const float3& v, float w) {
    dst.push_back(v.x);
    dst.push_back(v.y);
    dst.push_back(v.z);
    dst.push_back(w);
}

float bitsToFloat(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

}  // namespace

struct PtrDeviceScene {
    int device = 0;
    DeviceBuffer<uint4> qnodes;
    DeviceBuffer<uint4> wnodes;   // PTR_WIDE_NODES=1: four-wide nodes (SceneView::wnodes)
    DeviceBuffer<float4> nodes, tris, triNormals, spheres, materials, rects, rectLights, envRgba;
    DeviceBuffer<uint2> sphereInfo;
    DeviceBuffer<int32_t> lightIndexByRect;
    DeviceBuffer<float2> envCond, envMarg, cone;
    DeviceBuffer<float4> triUv, triTangent, texels, materialTex;
    DeviceBuffer<uint4> texInfo;
    DeviceBuffer<float> envPdf;
    SceneView view{};
    uint64_t info[8] = {0};
    double uploadSeconds = 0.0;
    double timings[4] = {0.0, 0.0, 0.0, 0.0};   // geometry preparation (or cache read), shading tables, copies to the device, 1 = geometry came from a cache
    uint64_t deviceTotalBytes = 0;        // hipDeviceProp_t::totalGlobalMem
    bool hasRandomWalkMaterial = false;   // a type-5 material with sssParams.y >= 0.5 (Metal random-walk subsurface)

    // render-time resources, grown on demand and kept across calls
    DeviceBuffer<float4> state, recBuf, itemAccum;   // state: the four 16 B words of every slot (PathPool::ray0 / ray1 / thr / accum)
    DeviceBuffer<float2> hit;
    DeviceBuffer<uint32_t> flushItem, signature, tailList, tailWords;
    DeviceBuffer<uint32_t> connectList, connectCounts;   // PathPool::connectList: per group a list and two sets of sub-list counters
    DeviceBuffer<uint32_t> busyLists, busyCounts;        // PathPool::busyIn / busyOut: per group two lists and three sets of counters
    // end of the frame: once the item queue is dry and at most this many slots are still alive, the remaining paths are finished by
    // k_tail_run (one lane per path, no launches between bounces) instead of further extend / shade / connect rounds; 0 = never
    uint64_t tailBelow = 512ull << 10;
    uint64_t poolSlots = 32ull << 20;        // resident path slots at most (PTR_POOL_SLOTS)
    // The pool is split into this many independent groups.  Two by default, each on a main stream (k_extend, k_shade) and a side stream
    // on which the k_connect of an iteration runs beside the k_extend of the next (they share nothing: one reads the rays k_shade wrote,
    // the other its connection records): four streams in flight, which is what the runtime's four hardware queues carry without
    // serialising.  (Rounds 1-3 ran four groups of one stream each; profiles/r3_ab_connect_overlap.txt.)
    // Frames of a few milliseconds (a pool of at most 8 Mi slots: config 1) keep four groups of one stream each: -6 % with two.
    uint32_t poolGroups = 0;      // PTR_POOL_GROUPS (0: two groups, four for small pools)
    bool connectOverlap = true;   // PTR_CONNECT_OVERLAP=0: k_connect on the group's own stream
    uint32_t maxPoolGroups() const { return poolGroups ? poolGroups : 4u; }
    uint32_t feederChunk = 256;   // slots per work-head claim while the pool is full (it grows as the pool drains)
    std::vector<hipStream_t> groupStreams;   // streams of groups 1.. (group 0 runs on the caller's stream)
    std::vector<hipEvent_t> groupEvents;
    std::vector<hipStream_t> sideStreams;    // per group: the stream of its k_connect launches (see poolGroups)
    std::vector<hipEvent_t> sideEvents;      // per group: k_shade of the iteration done / k_connect of the iteration done
    int refillBelow = 40;
    uint32_t spillLevels = 0;   // stack levels beyond the LDS part that the scene's tree can need (sizes the spill area)
    DeviceBuffer<uint4> medium;
    DeviceBuffer<uint32_t> scalars, pixelOfLocal, spill;
    DeviceBuffer<uint2> itemReserve;
    DeviceBuffer<uint32_t> itemHeads, zeros;
    DeviceBuffer<uint64_t> counters;
    DeviceBuffer<float> outBands;
    DeviceBuffer<float4> rayBatch;
    DeviceBuffer<PtrHit> hitBatch;
    uint32_t* pinnedAlive = nullptr;
    uint32_t traceGrid = 0;       // persistent blocks of the traversal kernels: fills every wave slot (also sizes the spill area)
    uint32_t traceGridHalf = 0;   // ... of k_extend / k_connect when several pool groups run large launches side by side
    // cached partition
    uint32_t cachedW = 0, cachedH = 0, cachedPart = 0, cachedParts = 0, cachedLocalPixels = 0;

    ~PtrDeviceScene() {
        if (pinnedAlive) (void)hipHostFree(pinnedAlive);
        for (hipStream_t st : groupStreams) (void)hipStreamDestroy(st);
        for (hipEvent_t e : groupEvents) (void)hipEventDestroy(e);
        for (hipStream_t st : sideStreams) (void)hipStreamDestroy(st);
        for (hipEvent_t e : sideEvents) (void)hipEventDestroy(e);
    }
};

namespace {

// per-group scalars: [0] unused, [1] k_extend work head, [2] k_connect work head, [3] pad,
//                    [4..19] ring of live-slot counters (one per iteration, written by k_extend at the end of a frame)
constexpr uint32_t kAliveRing = 16;
constexpr uint32_t kAliveBase = 4;
constexpr uint32_t kScalarCount = kAliveBase + kAliveRing;
constexpr uint64_t kHalfGridGroupSlots = 3ull << 20;   // groups at least this large launch k_extend / k_connect on half the wave slots
constexpr uint32_t kMaxPoolGroups = 8;        // one block of scalars / one spill area per group
constexpr uint32_t kPinnedHeadsOffset = 16;
constexpr uint32_t kTexInfoWords = 20;   // kernels/texture.h kTexInfoVec4 uint4 per texture   // pinned staging: [0..15] per-group live-slot counts, then kItemHeads range heads

// Stack spill area of one pool group: the stack levels beyond kLdsStackLevels, one column per thread of the persistent grid.  The
// four-wide walk pushes at most three entries per step (two binary levels), plus the root beside an oversize leaf: a tree of binary
// depth d needs 3 ceil(d / 2) + 2 entries, not the worst case kTraversalStackDepth (1 GB of HBM per scene for 8 groups).
size_t spillWordsPerGroup(const PtrDeviceScene& ds) { return static_cast<size_t>(ds.spillLevels) * ds.traceGrid * kTraceGridUnit; }

// 576 B MaterialData -> the 13 float4 the integrator reads (kernels/device_types.h MaterialSlot).
void compactMaterial(const PtrMaterial& m, std::vector<float>& out) {
    const float* rows[kMaterialVec4] = {m.baseColorRoughness, m.typeEta,        m.emission,           m.conductorEta,
                                        m.conductorK,         m.coatParams,     m.coatTint,           m.coatAbsorption,
                                        m.carpaintBaseParams, m.carpaintFlakeParams, m.carpaintBaseEta, m.carpaintBaseK,
                                        m.dielectricSigmaA,   m.sssSigmaA,      m.sssSigmaS,          m.sssParams};
    for (uint32_t r = 0; r < kMaterialVec4; ++r) {
        float v[4] = {rows[r][0], rows[r][1], rows[r][2], rows[r][3]};
        if (r == kMatCoatTint) v[3] = m.pbrParams[0];   // PBR metallic rides in the free w lane
        if (r == kMatDielectricSigmaA) v[3] = m.pbrExtras[2];   // PBR transmission factor (KHR_materials_transmission), Metal PBR model only
        out.insert(out.end(), v, v + 4);
    }
}

// Device-independent half of a scene upload: geometry bake + BVH, compact materials, light list, environment tables.  Built once
// and uploaded to every device a frame is rendered on (ptr_render_multi).
struct PreparedScene {
    ptr::PreparedGeometry pg;   // BVH, leaf-order arrays, four-wide nodes, node format: what a geometry cache file holds
    std::vector<float> mats, lights;
    std::vector<int32_t> lightIndexByRect;
    uint32_t lightCount = 0;
    bool lightsHaveTriangles = true;   // every rectangle light found its two triangles in the geometry (always, unless degenerate)
    bool hasRandomWalkMaterial = false;
    ptr::EnvImportanceDistribution envDist;
    bool hasEnvDist = false;
    // material textures (kernels/texture.h): every level of every texture in one array, the per-texture records, and the
    // per-material texture records; empty when the scene has no textures
    std::vector<float> texels;
    std::vector<uint32_t> texInfo;
    std::vector<float> materialTex;
    double geometrySeconds = 0.0;   // bake + BVH + leaf order + wide nodes, or reading them from a geometry cache
    double shadingSeconds = 0.0;    // materials, lights, environment tables, texture mips
    bool geometryFromCache = false;
    double seconds = 0.0;           // both
};

// Mip chain of one texture appended to `texels`: level l + 1 halves both sizes (at least 1) and averages the 2x2 block of level
// l under each of its texels, the second tap clamped at the edge of odd-sized levels; sums in the order ((a + b) + (c + d)) * 0.25.
void appendTextureWithMips(const PtrTexture& t, std::vector<float>& texels, std::vector<uint32_t>& info) {
    uint32_t w = t.width, h = t.height, levels = 1;
    for (uint32_t a = w, b = h; a > 1u || b > 1u; a = std::max(a / 2u, 1u), b = std::max(b / 2u, 1u)) ++levels;
    levels = std::min(levels, 16u);
    const size_t head = info.size();
    info.resize(head + kTexInfoWords, 0u);
    info[head + 0] = t.width;
    info[head + 1] = t.height;
    info[head + 2] = levels;
    info[head + 3] = (t.wrapS & 3u) | ((t.wrapT & 3u) << 2) | ((t.filter ? 1u : 0u) << 4);
    size_t prev = texels.size() / 4u;
    info[head + 4] = static_cast<uint32_t>(prev);
    texels.insert(texels.end(), t.rgba, t.rgba + static_cast<size_t>(w) * h * 4u);
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t nw = std::max(w / 2u, 1u), nh = std::max(h / 2u, 1u);
        const size_t at = texels.size() / 4u;
        info[head + 4 + l] = static_cast<uint32_t>(at);
        texels.resize(texels.size() + static_cast<size_t>(nw) * nh * 4u);
        for (uint32_t y = 0; y < nh; ++y) {
            const uint32_t y0 = std::min(2u * y, h - 1u), y1 = std::min(2u * y + 1u, h - 1u);
            for (uint32_t x = 0; x < nw; ++x) {
                const uint32_t x0 = std::min(2u * x, w - 1u), x1 = std::min(2u * x + 1u, w - 1u);
                const float* a = &texels[(prev + static_cast<size_t>(y0) * w + x0) * 4u];
                const float* b = &texels[(prev + static_cast<size_t>(y0) * w + x1) * 4u];
                const float* c = &texels[(prev + static_cast<size_t>(y1) * w + x0) * 4u];
                const float* d = &texels[(prev + static_cast<size_t>(y1) * w + x1) * 4u];
                float* o = &texels[(at + static_cast<size_t>(y) * nw + x) * 4u];
                for (int ch = 0; ch < 4; ++ch) o[ch] = ((a[ch] + b[ch]) + (c[ch] + d[ch])) * 0.25f;
            }
        }
        prev = at;
        w = nw;
        h = nh;
    }
}

// The geometry half of the preparation: everything a geometry cache file holds (host/geometry_cache.h).
void prepareGeometry(const PtrSceneDesc& desc, ptr::PreparedGeometry& pg) {
    const ptr::Knobs knobs = ptr::readKnobs();
    std::string geoError;
    if (!ptr::BuildSceneGeometry(desc, 0, pg.geo, geoError)) throw HipError{geoError};
    // 32 B quantised nodes halve the node fetches; use them unless the 16-bit grid is coarse next to the primitives (cell > 1/8 of
    // the mean primitive extent would inflate leaf boxes noticeably)
    const ptr::FlatBvh& bvh = pg.geo.bvh;
    const float maxCell = std::max(std::max(bvh.gridCell[0], bvh.gridCell[1]), bvh.gridCell[2]);
    pg.useQuantized = bvh.nodeCount > 0 && maxCell * 8.0f <= bvh.meanPrimExtent;
    if (knobs.quantizedNodes >= 0) pg.useQuantized = bvh.nodeCount > 0 && knobs.quantizedNodes != 0;
    // four-wide nodes for the persistent traversal kernels; the binary array stays for the cold kernels and the counting build
    if (pg.useQuantized && knobs.wideNodes != 0) {
        // (a tree so lopsided that the by-area wide tree would outgrow the traversal stack keeps the by-level collapse, whose depth is half
        // the binary tree's)
        pg.wideCount = ptr::BuildWideNodes(bvh, knobs.wideNodes == 2 ? ptr::WideCollapse::ByLevel : ptr::WideCollapse::ByArea, pg.wide, &pg.wideDepth);
        if (3u * pg.wideDepth + 4u > kTraversalStackDepth) pg.wideCount = ptr::BuildWideNodes(bvh, ptr::WideCollapse::ByLevel, pg.wide, &pg.wideDepth);
        if (static_cast<uint64_t>(pg.wideCount) * 64u > 0xFFFFFFFFull) throw HipError{"scene exceeds the 4 GiB node array limit"};
    }
}

// cachePath (may be null): read the geometry from that file instead of building it
void prepareScene(const PtrSceneDesc& desc, PreparedScene& ps, const char* cachePath = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    if (cachePath && *cachePath) {
        std::string error;
        if (!ptr::ReadGeometryCache(cachePath, ptr::SceneFingerprint(desc), ps.pg, error)) throw HipError{error};
        ps.geometryFromCache = true;
    } else {
        prepareGeometry(desc, ps.pg);
    }
    ps.geometrySeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const ptr::SceneGeometry& geoRef = ps.pg.geo;

    // compact materials
    ps.mats.reserve(static_cast<size_t>(desc.materialCount) * kMaterialVec4 * 4);
    for (uint32_t i = 0; i < desc.materialCount; ++i) compactMaterial(desc.materials[i], ps.mats);
    for (uint32_t i = 0; i < desc.materialCount; ++i) {
        const PtrMaterial& m = desc.materials[i];
        if (static_cast<uint32_t>(m.typeEta[0]) == PTR_MAT_SUBSURFACE && m.sssParams[1] >= 0.5f) ps.hasRandomWalkMaterial = true;
    }

    // rectangle lights: DiffuseLight rectangles with non-zero emission (:2484-2522)
    ps.lightIndexByRect.assign(desc.rectCount, -1);
    if (desc.rectCount > 0 && desc.materialCount > 0) {
        for (uint32_t i = 0; i < desc.rectCount; ++i) {
            const PtrRect& r = desc.rects[i];
            const PtrMaterial& m = desc.materials[std::min(r.materialTwoSided[0], desc.materialCount - 1u)];
            if (static_cast<uint32_t>(m.typeEta[0]) != PTR_MAT_DIFFUSE_LIGHT) continue;
            const float3 e{m.emission[0], m.emission[1], m.emission[2]};
            if (!(ptr::dot(e, e) > 0.0f)) continue;
            const float3 c{r.corner[0], r.corner[1], r.corner[2]}, eu{r.edgeU[0], r.edgeU[1], r.edgeU[2]},
                ev{r.edgeV[0], r.edgeV[1], r.edgeV[2]};
            const float3 n = ptr::normalize(float3{r.normalAndPlane[0], r.normalAndPlane[1], r.normalAndPlane[2]});
            put4(ps.lights, c, ptr::length(ptr::cross(eu, ev)));
            put4(ps.lights, eu, r.materialTwoSided[1] != 0u ? 1.0f : 0.0f);
            put4(ps.lights, ev, bitsToFloat(i));
            // rows 5..10: the light's own two triangles exactly as the traversal reads them (leaf-order records), for k_shade's
            // self-occlusion test of a light sample; normal.w = 1 when they are present
            const uint32_t t0 = geoRef.rectTriLeaf[static_cast<size_t>(i) * 2u], t1 = geoRef.rectTriLeaf[static_cast<size_t>(i) * 2u + 1u];
            const bool haveTris = t0 != 0xFFFFFFFFu && t1 != 0xFFFFFFFFu;
            put4(ps.lights, n, haveTris ? 1.0f : 0.0f);
            ps.lightsHaveTriangles = ps.lightsHaveTriangles && haveTris;
            put4(ps.lights, e, 0.0f);
            for (uint32_t t : {t0, t1}) {
                for (int row = 0; row < 3; ++row) {
                    const float* src = haveTris ? &geoRef.triData[static_cast<size_t>(t) * 12u + static_cast<size_t>(row) * 4u] : nullptr;
                    for (int c = 0; c < 4; ++c) ps.lights.push_back(src ? src[c] : 0.0f);
                }
            }
            ps.lightIndexByRect[i] = static_cast<int32_t>(ps.lightCount++);
        }
    }
    if (desc.envRgba && desc.envWidth > 0 && desc.envHeight > 0) {
        ps.hasEnvDist = ptr::BuildEnvImportanceDistribution(desc.envRgba, desc.envWidth, desc.envHeight, &ps.envDist);
    }
    if (desc.textures && desc.textureCount > 0 && !geoRef.triUv.empty()) {
        for (uint32_t i = 0; i < desc.textureCount; ++i) {
            if (!desc.textures[i].rgba || desc.textures[i].width == 0 || desc.textures[i].height == 0) throw HipError{"texture without pixels"};
            appendTextureWithMips(desc.textures[i], ps.texels, ps.texInfo);
            if (ps.texels.size() / 4u > 0xFFFFFFF0ull) throw HipError{"textures exceed the 4 G texel limit"};
        }
        ps.materialTex.assign(static_cast<size_t>(desc.materialCount) * kMaterialTexVec4 * 4u, 0.0f);
        for (uint32_t i = 0; i < desc.materialCount; ++i) {
            const PtrMaterial& m = desc.materials[i];
            float* o = &ps.materialTex[static_cast<size_t>(i) * kMaterialTexVec4 * 4u];
            for (int r = 0; r < 12; ++r) std::memcpy(o + r * 4, m.textureTransform[r], 16);
            for (int k = 0; k < 4; ++k) o[48 + k] = bitsToFloat(m.textureIndices0[k]);
            o[52] = bitsToFloat(m.textureIndices1[0]);
            o[53] = bitsToFloat(m.textureIndices1[1]);
            const uint32_t uvSets = (std::min(m.textureUvSet0[0], 1u) << 0) | (std::min(m.textureUvSet0[1], 1u) << 1) | (std::min(m.textureUvSet0[2], 1u) << 2) |
                                    (std::min(m.textureUvSet0[3], 1u) << 3) | (std::min(m.textureUvSet1[0], 1u) << 4) | (std::min(m.textureUvSet1[1], 1u) << 5);
            o[54] = bitsToFloat(uvSets);
            o[55] = bitsToFloat(m.materialFlags);
            std::memcpy(o + 56, m.pbrParams, 16);
            std::memcpy(o + 60, m.pbrExtras, 16);
        }
    }
    ps.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ps.shadingSeconds = ps.seconds - ps.geometrySeconds;
}

void uploadScene(const PtrSceneDesc& desc, const PreparedScene& ps, PtrDeviceScene& ds) {
    const auto t0 = std::chrono::steady_clock::now();
    const ptr::SceneGeometry& geo = ps.pg.geo;
    const ptr::FlatBvh& bvh = geo.bvh;
    const uint32_t triCount = geo.triCount;
    const uint32_t lightCount = ps.lightCount;
    ds.hasRandomWalkMaterial = ps.hasRandomWalkMaterial;

    HIP_CHECK(hipSetDevice(ds.device));
    // only the node array the kernels will read goes to the device (a 29 M-triangle scene: 0.5 GB instead of 1.5 GB of nodes)
    const ptr::Knobs knobs = ptr::readKnobs();
    const bool useQuantized = ps.pg.useQuantized;
    if (useQuantized) {
        ds.nodes.release();
        ds.qnodes.upload(reinterpret_cast<const uint4*>(bvh.qnodes.data()), bvh.qnodes.size() / 4);
    } else {
        ds.qnodes.release();
        ds.nodes.upload(reinterpret_cast<const float4*>(bvh.nodes.data()), bvh.nodes.size() / 4);
    }
    ds.tris.upload(reinterpret_cast<const float4*>(geo.triData.data()), geo.triData.size() / 4);
    ds.triNormals.upload(reinterpret_cast<const float4*>(geo.triNormals.data()), geo.triNormals.size() / 4);
    ds.spheres.upload(reinterpret_cast<const float4*>(geo.sphereData.data()), geo.sphereData.size() / 4);
    ds.sphereInfo.upload(reinterpret_cast<const uint2*>(geo.sphereInfo.data()), geo.sphereInfo.size() / 2);
    ds.materials.upload(reinterpret_cast<const float4*>(ps.mats.data()), ps.mats.size() / 4);
    ds.rects.upload(reinterpret_cast<const float4*>(desc.rects), static_cast<size_t>(desc.rectCount) * 5);
    ds.rectLights.upload(reinterpret_cast<const float4*>(ps.lights.data()), ps.lights.size() / 4);
    ds.lightIndexByRect.upload(ps.lightIndexByRect.data(), ps.lightIndexByRect.size());

    SceneView& v = ds.view;
    std::memset(&v, 0, sizeof(v));
    v.nodes = ds.nodes.ptr;
    v.tris = ds.tris.ptr;
    v.triNormals = ds.triNormals.ptr;
    v.spheres = ds.spheres.ptr;
    v.sphereInfo = ds.sphereInfo.ptr;
    v.materials = ds.materials.ptr;
    v.rects = ds.rects.ptr;
    v.rectLights = ds.rectLights.ptr;
    v.lightIndexByRect = ds.lightIndexByRect.ptr;
    v.qnodes = ds.qnodes.ptr;
    std::memcpy(v.gridOrigin, bvh.gridOrigin, sizeof(v.gridOrigin));
    std::memcpy(v.gridCell, bvh.gridCell, sizeof(v.gridCell));
    for (int a = 0; a < 3; ++a) v.gridInvCell[a] = 1.0f / bvh.gridCell[a];
    v.useQuantized = useQuantized ? 1u : 0u;
    if (ps.pg.wideCount > 0u) {
        ds.wnodes.upload(reinterpret_cast<const uint4*>(ps.pg.wide.get()), static_cast<size_t>(ps.pg.wideCount) * 4u);
        v.wnodes = ds.wnodes.ptr;
        v.wideBytes = static_cast<uint32_t>(static_cast<size_t>(ps.pg.wideCount) * 64u);
        v.useWide = 1u;
    }
    const size_t nodeBytes = v.useQuantized ? bvh.qnodes.size() * 4u : bvh.nodes.size() * 4u;
    const size_t triBytes = geo.triData.size() * 4u;
    if (nodeBytes > 0xFFFFFFFFull || triBytes > 0xFFFFFFFFull) throw HipError{"scene exceeds the 4 GiB node/triangle array limit"};
    v.nodeBytes = static_cast<uint32_t>(nodeBytes);
    v.triBytes = static_cast<uint32_t>(triBytes);
    v.rootRef = bvh.rootRef;
    v.oversizeRef = bvh.oversizeRef;
    v.materialCount = desc.materialCount;
    v.rectCount = desc.rectCount;
    v.rectLightCount = lightCount;
    for (uint32_t i = 0; i < desc.materialCount; ++i) v.materialTypes |= 1u << std::min(static_cast<uint32_t>(desc.materials[i].typeEta[0]), 7u);
    v.settleRectLights = (lightCount > 0u && lightCount <= 8u && ps.lightsHaveTriangles) ? 1u : 0u;   // kSettleLightsMax of wavefront.hip

    if (desc.envRgba && desc.envWidth > 0 && desc.envHeight > 0) {
        const size_t texels = static_cast<size_t>(desc.envWidth) * desc.envHeight;
        ds.envRgba.upload(reinterpret_cast<const float4*>(desc.envRgba), texels);
        v.envRgba = ds.envRgba.ptr;
        v.envWidth = desc.envWidth;
        v.envHeight = desc.envHeight;
        if (ps.hasEnvDist) {
            const ptr::EnvImportanceDistribution& dist = ps.envDist;
            static_assert(sizeof(ptr::AliasEntry) == sizeof(float2), "alias entry layout");
            ds.envCond.upload(reinterpret_cast<const float2*>(dist.conditional.data()), dist.conditional.size());
            ds.envMarg.upload(reinterpret_cast<const float2*>(dist.marginal.data()), dist.marginal.size());
            ds.envPdf.upload(dist.texelPdf.data(), dist.texelPdf.size());
            v.envCond = ds.envCond.ptr;
            v.envMarg = ds.envMarg.ptr;
            v.envPdf = ds.envPdf.ptr;
            v.envSampling = 1u;
        }
    }

    if (!ps.texels.empty()) {
        ds.triUv.upload(reinterpret_cast<const float4*>(geo.triUv.data()), geo.triUv.size() / 4);
        ds.triTangent.upload(reinterpret_cast<const float4*>(geo.triTangent.data()), geo.triTangent.size() / 4);
        ds.texels.upload(reinterpret_cast<const float4*>(ps.texels.data()), ps.texels.size() / 4);
        ds.texInfo.upload(reinterpret_cast<const uint4*>(ps.texInfo.data()), ps.texInfo.size() / 4);
        ds.materialTex.upload(reinterpret_cast<const float4*>(ps.materialTex.data()), ps.materialTex.size() / 4);
        v.triUv = ds.triUv.ptr;
        v.triTangent = ds.triTangent.ptr;
        v.texels = ds.texels.ptr;
        v.texInfo = ds.texInfo.ptr;
        v.materialTex = ds.materialTex.ptr;
        v.textureCount = desc.textureCount;
    }

    ds.info[0] = bvh.nodeCount;
    ds.info[1] = bvh.leafCount;
    ds.info[2] = triCount;
    ds.info[3] = desc.sphereCount;
    ds.info[4] = bvh.maxDepth;
    ds.info[5] = bvh.maxLeafSize;
    ds.info[6] = static_cast<uint64_t>(bvh.sahCost * 1000.0);
    ds.info[7] = lightCount;

    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, ds.device));
    ds.deviceTotalBytes = prop.totalGlobalMem;
    const uint32_t cus = prop.multiProcessorCount > 0 ? static_cast<uint32_t>(prop.multiProcessorCount) : 256u;
    ds.traceGrid = cus * 8u;   // 8 blocks of 256 threads per CU: fills the wave slots, grid-stride the rest
    if (knobs.refillBelow > 0) ds.refillBelow = knobs.refillBelow;
    // Half the wave slots when the pool runs as several groups of large launches: the kernels of the other groups (k_shade above all,
    // which needs 128 VGPRs a wave) then always find room beside a traversal kernel instead of queueing behind its last waves, and each
    // persistent wave sees twice as many rays before its own tail.  Measured on configs 2 / 3 / 4: +3.5 / +1.5 / +1.5 %, one rank of
    // eight 47.7 -> 44.5 ms; frames of a few milliseconds (config 1) lose 10 % and keep the full grid (profiles/r2_ab_grid_pool_knobs.txt).
    ds.traceGridHalf = cus * 4u;
    if (knobs.tailBelow >= 0) ds.tailBelow = static_cast<uint64_t>(knobs.tailBelow);
    if (knobs.poolSlots != 0) ds.poolSlots = knobs.poolSlots;
    if (knobs.poolGroups != 0) ds.poolGroups = knobs.poolGroups;
    ds.connectOverlap = knobs.connectOverlap != 0;
    // stack entries a ray of this tree can need: one per binary level for the two-box walk, three per level of the wide tree for the
    // four-wide walk (prepareGeometry keeps 3 x depth + 4 within the stack), the root beside an oversize leaf, and a margin
    const uint32_t wideLevels = ps.pg.wideCount > 0u ? ps.pg.wideDepth : 0u;
    const uint32_t stackNeed = std::min<uint32_t>(kTraversalStackDepth, std::max(3u * wideLevels, static_cast<uint32_t>(bvh.maxDepth)) + 4u);
    v.stackLimit = std::max(stackNeed, kLdsStackLevels);
    ds.spillLevels = v.stackLimit - kLdsStackLevels;
    ds.spill.ensure(std::max<size_t>(spillWordsPerGroup(ds) * ds.maxPoolGroups() * 2u, 1u));   // a group's k_extend and k_connect may run side by side: an area each
    ds.scalars.ensure(static_cast<size_t>(kScalarCount) * kMaxPoolGroups);
    ds.counters.ensure(kCounterSlots);
    ds.zeros.ensure(16);
    HIP_CHECK(hipMemset(ds.zeros.ptr, 0, 16 * sizeof(uint32_t)));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ds.pinnedAlive), sizeof(uint32_t) * (kPinnedHeadsOffset + kItemHeads), hipHostMallocDefault));
    const double copySeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ds.uploadSeconds = ps.seconds + copySeconds;
    ds.timings[0] = ps.geometrySeconds;
    ds.timings[1] = ps.shadingSeconds;
    ds.timings[2] = copySeconds;
    ds.timings[3] = ps.geometryFromCache ? 1.0 : 0.0;
    if (knobs.verboseBuild) std::fprintf(stderr, "[upload] prepare %.2f s, copies to the device %.2f s\n", ps.seconds, copySeconds);
}

void buildScene(const PtrSceneDesc& desc, PtrDeviceScene& ds, const char* cachePath = nullptr) {
    PreparedScene ps;
    prepareScene(desc, ps, cachePath);
    uploadScene(desc, ps, ds);
}

// Camera basis on the host (BuildCamera, EmbreeHeadlessRenderer.mm:150-198).
void buildCamera(const PtrSettings& s, CameraParams& c) {
    constexpr float kPiF = 3.14159265358979323846f;
    const float aspect = s.width > 0 ? static_cast<float>(s.width) / static_cast<float>(s.height) : 1.0f;
    const float vfov = std::min(std::max(s.cameraVerticalFov, 1.0f), 179.0f);
    const float defocus = std::max(s.cameraDefocusAngle, 0.0f);
    const float theta = vfov * (kPiF / 180.0f);
    const float h = std::tan(theta * 0.5f);
    const float viewportHeight = 2.0f * h;
    const float viewportWidth = aspect * viewportHeight;
    const float distance = std::max(s.cameraDistance, 0.1f);
    const float cp = std::cos(s.cameraPitch), sp = std::sin(s.cameraPitch);
    const float cy = std::cos(s.cameraYaw), sy = std::sin(s.cameraYaw);
    const float3 offset{distance * cp * cy, distance * sp, distance * cp * sy};
    const float3 lookAt{s.cameraTarget[0], s.cameraTarget[1], s.cameraTarget[2]};
    const float3 lookFrom = lookAt + offset;
    const float3 w = ptr::normalize(lookFrom - lookAt);
    const float3 u = ptr::normalize(ptr::cross(float3{0.0f, 1.0f, 0.0f}, w));
    const float3 v = ptr::cross(w, u);
    float focus = s.cameraFocusDistance;
    if (focus <= 0.0f) focus = distance;
    const float3 horizontal = (focus * viewportWidth) * u;
    const float3 vertical = (focus * viewportHeight) * v;
    const float3 lowerLeft = ((lookFrom - 0.5f * horizontal) - 0.5f * vertical) - focus * w;
    auto st = [](float* d, const float3& a) {
        d[0] = a.x;
        d[1] = a.y;
        d[2] = a.z;
    };
    st(c.origin, lookFrom);
    st(c.lowerLeft, lowerLeft);
    st(c.horizontal, horizontal);
    st(c.vertical, vertical);
    st(c.u, u);
    st(c.v, v);
    c.lensRadius = focus * std::tan((defocus * 0.5f) * (kPiF / 180.0f));
}

void fillRenderParams(const PtrSettings& s, uint32_t spp, RenderParams& rp) {
    std::memset(&rp, 0, sizeof(rp));
    buildCamera(s, rp.cam);
    rp.width = s.width;
    rp.byWidth = makeDivU32(s.width);
    rp.height = s.height;
    rp.maxDepth = std::min(s.maxDepth, kFlagFieldMask);
    rp.shadowSlack = (s.debugShadowSlack > 0.0f && s.debugShadowSlack < 1.0f) ? s.debugShadowSlack : 0.0f;
    rp.seedBase = s.seed != 0 ? s.seed : 0x9e3779b9u;
    rp.spp = std::max(1u, spp);
    rp.sampleBase = 0u;
    rp.sppTotal = rp.spp;
    rp.passFlags = 3u;   // one pass: first and last
    rp.enableRussianRoulette = s.enableRussianRoulette;
    rp.enableSpecularNee = s.enableSpecularNee;
    rp.enableMnee = s.enableMnee;
    rp.enableMneeSecondary = s.enableMneeSecondary;
    rp.backgroundMode = s.backgroundMode;
    std::memcpy(rp.backgroundColor, s.backgroundColor, sizeof(rp.backgroundColor));
    rp.envRotation = s.environmentRotation;
    rp.envIntensity = s.environmentIntensity;
    rp.clampFactor = std::max(s.fireflyClampFactor, 0.0f);   // MakeFireflyParams, :381-391
    rp.clampFloor = std::max(s.fireflyClampFloor, 0.0f);
    rp.throughputClamp = std::max(s.throughputClamp, 0.0f);
    rp.tailClampBase = std::max(s.specularTailClampBase, 0.0f);
    rp.tailClampRoughnessScale = std::max(s.specularTailClampRoughnessScale, 0.0f);
    rp.minSpecularPdf = std::max(s.minSpecularPdf, 1.0e-8f);
    rp.clampEnabled = s.fireflyClampEnabled ? 1.0f : 0.0f;
    rp.emissionScale = (s.emissionScale > 0.0f && std::isfinite(s.emissionScale)) ? s.emissionScale : 1.0f;
    rp.mediaMode = s.metalSemantics & (PTR_METAL_MEDIA | PTR_METAL_THIN | PTR_METAL_FACE_NORMAL | PTR_METAL_SPECULAR | PTR_METAL_SSS | PTR_METAL_PBR |
                                       PTR_METAL_CLAMPS);
    rp.clampMaxContribution = std::max(s.fireflyClampMaxContribution, 0.0f);   // make_firefly_params, pathtrace.metal:3545
    rp.minSpecularPdfRaw = s.minSpecularPdf;
    rp.sssMode = s.sssMode;
    rp.sssMaxSteps = std::max(s.sssMaxSteps, 1u);
}

// Local pixel order of a partition: its PTR_BAND_ROWS-row bands top to bottom, each walked in 8x8 blocks so the
// 64 lanes of a wave start with a compact, coherent bundle of primary rays.
void partitionPixels(uint32_t width, uint32_t height, uint32_t part, uint32_t parts, std::vector<uint32_t>& out) {
    out.clear();
    const uint32_t bands = (height + PTR_BAND_ROWS - 1u) / PTR_BAND_ROWS;
    for (uint32_t b = part; b < bands; b += parts) {
        const uint32_t y0 = b * PTR_BAND_ROWS, y1 = std::min(y0 + PTR_BAND_ROWS, height);
        for (uint32_t ty = y0; ty < y1; ty += 8u) {
            for (uint32_t tx = 0; tx < width; tx += 8u) {
                for (uint32_t y = ty; y < std::min(ty + 8u, y1); ++y) {
                    for (uint32_t x = tx; x < std::min(tx + 8u, width); ++x) out.push_back(y * width + x);
                }
            }
        }
    }
}

struct EventTimer {
    std::vector<hipEvent_t> events;
    size_t used = 0;
    ~EventTimer() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
    hipEvent_t next() {
        if (used == events.size()) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            events.push_back(e);
        }
        return events[used++];
    }
};

// Bytes the per-sample accumulators of one pass may take: a quarter of the device's TOTAL memory, at most 16 GiB.  Taken
// from the device's size, never from what happens to be free: every rank of a multi-GPU render has to split a frame
// into the same passes (k_resolve adds the per-pass sums in pass order), whatever else lives on its card.
uint64_t itemBudgetBytes(const PtrDeviceScene& ds) {
    return std::max<uint64_t>(64ull << 20, std::min<uint64_t>(16ull << 30, ds.deviceTotalBytes / 4u));
}

// One pass over `spp` samples per pixel starting at sample `sampleBase` of a frame of `sppTotal`; passFlags bit 0 = first pass of
// the frame (output and counters start from zero), bit 1 = last pass (the running sum in dOut is divided by sppTotal).
void renderPass(PtrDeviceScene& ds, const PtrSettings& settings, uint32_t spp, uint32_t sampleBase, uint32_t sppTotal, uint32_t passFlags,
                uint32_t part, uint32_t parts, float* dOut, hipStream_t stream, int mode, PtrRenderStats* stats) {
    const bool count = (mode & 1) != 0;         // counting instantiation of the kernels
    const bool soloGroup = (mode & 2) != 0;     // one pool group: kernels run alone, for clean per-kernel timings
    if (settings.width == 0 || settings.height == 0) throw HipError{"render size must be non-zero"};
    if (parts == 0 || part >= parts) throw HipError{"bad partition"};
    HIP_CHECK(hipSetDevice(ds.device));

    RenderParams rp;
    fillRenderParams(settings, spp, rp);
    rp.sampleBase = sampleBase;
    rp.sppTotal = std::max(1u, sppTotal);
    rp.passFlags = passFlags;

    if (ds.cachedW != settings.width || ds.cachedH != settings.height || ds.cachedPart != part || ds.cachedParts != parts) {
        std::vector<uint32_t> pixels;
        partitionPixels(settings.width, settings.height, part, parts, pixels);
        ds.pixelOfLocal.upload(pixels.data(), pixels.size());
        ds.cachedW = settings.width;
        ds.cachedH = settings.height;
        ds.cachedPart = part;
        ds.cachedParts = parts;
        ds.cachedLocalPixels = static_cast<uint32_t>(pixels.size());
    }
    const uint32_t localPixels = ds.cachedLocalPixels;
    const uint32_t bandCount = ptr_part_band_count(settings.height, part, parts);
    const size_t outFloats = static_cast<size_t>(bandCount) * PTR_BAND_ROWS * settings.width * 3u;
    if (passFlags & 1u) HIP_CHECK(hipMemsetAsync(dOut, 0, outFloats * sizeof(float), stream));
    if (localPixels == 0) {
        HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }

    // Work items = single pixel samples: item w = sample * localPixels + localPixel.  Slots claim items from 64 range heads,
    // so the pool stays full until the last items regardless of how path length varies over the image.  (Items of C > 1
    // consecutive samples lengthen the end of the frame - once the queue is dry every slot still finishes its item, C = 4
    // cost 12 % on config 2 - so a frame whose per-sample accumulators do not fit is rendered in passes instead, see
    // renderBands.)
    rp.localPixels = localPixels;
    rp.byLocalPixels = makeDivU32(localPixels);
    const uint64_t itemCount64 = static_cast<uint64_t>(localPixels) * rp.spp;
    if (itemCount64 > 0xFFFF0000ull) throw HipError{"too many work items for one pass (reduce spp or resolution)"};
    rp.itemCount = static_cast<uint32_t>(itemCount64);
    // enough to keep every CU's wave slots full several times over - but never more than half the work items: a pool as
    // large as the frame is all ramp-up and drain (config 1, 16.8 M samples: 16 Mi slots 16.2 ms, 8 Mi slots 10.0 ms)
    // (32 Mi slots at most: a larger pool gives every launch more rays before its tail - config 2 +4.5 %, config 4 +2 %, config 5 +8 %
    // against 16 Mi - and since the lists made the end of the frame cheap it costs the partitions of a multi-GPU frame nothing:
    // one rank of eight 38.2 ms at 16 Mi, 38.4 ms at 32 Mi; profiles/r2_ab_grid_pool_knobs.txt)
    const uint64_t targetSlots = std::min<uint64_t>(ds.poolSlots, std::max<uint64_t>(1ull << 20, itemCount64 / 2u));
    uint32_t slots = static_cast<uint32_t>(std::min<uint64_t>(targetSlots, itemCount64));
    if (slots < itemCount64) slots &= ~255u;   // item ranges start right after the pre-assigned items: keep them 64-aligned
    // items [0, slots) are pre-assigned by k_generate; the rest is split into kItemHeads ranges with one head each
    rp.itemHeadFirst = slots;
    rp.itemsPerHead = static_cast<uint32_t>(((itemCount64 - slots + kItemHeads - 1u) / kItemHeads + 63u) & ~63ull);

    ds.state.ensure(static_cast<size_t>(slots) * 4u);
    ds.hit.ensure(slots);
    ds.flushItem.ensure(slots);
    if (count) ds.signature.ensure(slots);
    if (ds.tailBelow > 0) {
        ds.tailList.ensure(slots);
        ds.tailWords.ensure(4);
    }
    ds.itemAccum.ensure(rp.itemCount);
    ds.recBuf.ensure(static_cast<size_t>(slots) * kRecSlots * 4u);
    ds.itemReserve.ensure((slots + 63u) / 64u);
    if (rp.mediaMode & PTR_METAL_MEDIA) ds.medium.ensure(slots);
    const bool texturedPaths = (rp.mediaMode & PTR_METAL_PBR) && ds.view.textureCount > 0u;   // the paths carry a ray cone
    if (texturedPaths) ds.cone.ensure(slots);

    PathPool pool;
    std::memset(&pool, 0, sizeof(pool));
    pool.ray0 = ds.state.ptr;
    pool.ray1 = ds.state.ptr + slots;
    pool.thr = ds.state.ptr + 2ull * slots;
    pool.accum = ds.state.ptr + 3ull * slots;
    pool.hit = ds.hit.ptr;
    pool.flushItem = ds.flushItem.ptr;
    pool.signature = count ? ds.signature.ptr : nullptr;
    pool.medium = (rp.mediaMode & PTR_METAL_MEDIA) ? ds.medium.ptr : nullptr;
    pool.cone = texturedPaths ? ds.cone.ptr : nullptr;
    pool.itemAccum = ds.itemAccum.ptr;
    ds.itemHeads.ensure(kItemHeadWords);
    pool.nextItem = ds.itemHeads.ptr;
    for (uint32_t k = 0; k < kRecSlots; ++k) {
        float4* base = ds.recBuf.ptr + static_cast<size_t>(k) * 4u * slots;
        pool.rec[k].org = base;
        pool.rec[k].dir = base + slots;
        pool.rec[k].a = base + 2ull * slots;
        pool.rec[k].b = base + 3ull * slots;
    }
    pool.itemReserve = ds.itemReserve.ptr;
    pool.pixelOfLocal = ds.pixelOfLocal.ptr;
    pool.counters = ds.counters.ptr;
    pool.zero = reinterpret_cast<const float4*>(ds.zeros.ptr);
    pool.slots = slots;
    pool.recStride = slots;

    // The pool is cut into independent groups, each driven through extend -> shade -> connect on its own HIP
    // stream.  Every launch of the persistent traversal kernels ends with a drain phase (the last rays of the
    // last chunks, few lanes busy); with two groups in flight the other group's kernels fill the CUs a draining
    // kernel leaves idle.  Groups share only the work-item head (one atomic per 64 items), so results do not
    // depend on how they interleave.
    struct Group {
        PathPool pool;
        LaunchConfig cfg;
        hipStream_t stream;
        uint32_t* scalars;
        bool done;
        uint32_t feederChunk;   // slots per work-head atomic; grows as the group drains at the end of the frame
        uint32_t* connectCounts = nullptr;   // two sets of sub-list counters, used in turn
        // busy lists (end of the frame): two lists and three counter sets in rotation.  busyStage 0: the kernels walk the slots;
        // 1: this iteration's k_shade (still walking the slots) fills the first list; 2: k_extend and k_shade walk the list of the
        // previous iteration and k_shade fills the next one
        uint32_t* busyLists = nullptr;
        uint32_t* busyCounts = nullptr;
        uint32_t busyStage = 0, busyTurn = 0;
        bool shadeListed = false;
    };
    const uint32_t wantGroups = ds.poolGroups ? ds.poolGroups : (slots > (8u << 20) ? 2u : 4u);
    uint32_t groupCount = std::min<uint32_t>(soloGroup ? 1u : wantGroups, std::max<uint32_t>(1u, slots >> 20));   // >= 1 Mi slots per group
    const uint32_t groupSlots = ((slots + groupCount - 1u) / groupCount + 255u) & ~255u;
    groupCount = (slots + groupSlots - 1u) / groupSlots;
    while (ds.groupStreams.size() + 1 < groupCount) {
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ds.groupStreams.push_back(st);
    }
    while (ds.groupEvents.size() < groupCount + 1) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ds.groupEvents.push_back(e);
    }
    const size_t spillWords = spillWordsPerGroup(ds);
    // k_connect beside the next k_extend: only while the streams fit the hardware queues (two groups at most), and never in a solo
    // render, whose point is kernels that do not overlap
    const bool overlap = ds.connectOverlap && !soloGroup && groupCount <= 2u;
    while (overlap && ds.sideStreams.size() < groupCount) {
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ds.sideStreams.push_back(st);
        for (int k = 0; k < 2; ++k) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ds.sideEvents.push_back(e);
        }
    }
    // connect lists: sub-list w % 64 takes the entries of k_shade's wave w, at most 64 each
    const uint32_t connectRegion = ((groupSlots + 63u) / 64u + kConnectQueues - 1u) / kConnectQueues * 64u;
    const size_t connectListWords = static_cast<size_t>(connectRegion) * kConnectQueues;
    const size_t connectCountWords = static_cast<size_t>(kConnectQueues) * kConnectCountStride;   // one set
    if (slots >= (1u << kConnectMaskShift)) throw HipError{"path-slot pool too large for the connect lists"};   // (the pool is capped at 64 Mi slots)
    ds.connectList.ensure(connectListWords * groupCount);
    ds.connectCounts.ensure(connectCountWords * 2u * kMaxPoolGroups);
    ds.busyLists.ensure(connectListWords * 2u * groupCount);
    ds.busyCounts.ensure(connectCountWords * 3u * kMaxPoolGroups);
    std::vector<Group> groups(groupCount);
    for (uint32_t g = 0; g < groupCount; ++g) {
        Group& gr = groups[g];
        const uint32_t first = g * groupSlots;
        gr.pool = pool;
        gr.pool.ray0 += first;
        gr.pool.ray1 += first;
        gr.pool.hit += first;
        gr.pool.thr += first;
        gr.pool.accum += first;
        gr.pool.flushItem += first;
        if (gr.pool.signature) gr.pool.signature += first;
        if (gr.pool.medium) gr.pool.medium += first;
        if (gr.pool.cone) gr.pool.cone += first;
        for (uint32_t k = 0; k < kRecSlots; ++k) {
            gr.pool.rec[k].org += first;
            gr.pool.rec[k].dir += first;
            gr.pool.rec[k].a += first;
            gr.pool.rec[k].b += first;
        }
        gr.pool.itemReserve += first / 64u;
        gr.pool.slots = std::min(groupSlots, slots - first);
        gr.pool.connectList = ds.connectList.ptr + connectListWords * g;
        gr.pool.connectRegion = connectRegion;
        gr.connectCounts = ds.connectCounts.ptr + connectCountWords * 2u * g;
        gr.busyLists = ds.busyLists.ptr + connectListWords * 2u * g;
        gr.busyCounts = ds.busyCounts.ptr + connectCountWords * 3u * g;
        gr.scalars = ds.scalars.ptr + static_cast<size_t>(g) * kScalarCount;
        const bool sideBySide = groupCount > 1 && gr.pool.slots >= kHalfGridGroupSlots;
        gr.cfg = LaunchConfig{sideBySide ? ds.traceGridHalf : ds.traceGrid, ds.spill.ptr + g * spillWords, gr.scalars + 1, ds.refillBelow};
        gr.stream = g == 0 ? stream : ds.groupStreams[g - 1];
        gr.done = false;
        gr.feederChunk = ds.feederChunk;
    }

    const bool timed = stats != nullptr;
    EventTimer timer;
    struct Span {
        hipEvent_t a, b;
        int kind;
        const void* stream;
    };
    std::vector<Span> spans;
    auto timedLaunch = [&](int kind, hipStream_t st, auto&& fn) {
        if (timed) {
            const hipEvent_t a = timer.next(), b = timer.next();
            HIP_CHECK(hipEventRecord(a, st));
            fn();
            HIP_CHECK(hipEventRecord(b, st));
            spans.push_back({a, b, kind, st});
        } else {
            fn();
        }
    };

    if (count && (passFlags & 1u)) HIP_CHECK(hipMemsetAsync(ds.counters.ptr, 0, sizeof(uint64_t) * kCounterSlots, stream));   // counters add up over the passes
    HIP_CHECK(hipMemsetAsync(ds.scalars.ptr, 0, sizeof(uint32_t) * kScalarCount * kMaxPoolGroups, stream));
    HIP_CHECK(hipMemsetAsync(ds.connectCounts.ptr, 0, sizeof(uint32_t) * connectCountWords * 2u * kMaxPoolGroups, stream));
    HIP_CHECK(hipMemsetAsync(ds.busyCounts.ptr, 0, sizeof(uint32_t) * connectCountWords * 3u * kMaxPoolGroups, stream));
    {
        uint32_t* heads = ds.pinnedAlive + kPinnedHeadsOffset;
        for (uint32_t k = 0; k < kItemHeads; ++k) {
            heads[k] = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(rp.itemHeadFirst) + static_cast<uint64_t>(k) * rp.itemsPerHead, rp.itemCount));
        }
        HIP_CHECK(hipMemsetAsync(pool.nextItem, 0, sizeof(uint32_t) * kItemHeadWords, stream));
        HIP_CHECK(hipMemcpy2DAsync(pool.nextItem, sizeof(uint32_t) * kItemHeadStride, heads, sizeof(uint32_t), sizeof(uint32_t), kItemHeads,
                                   hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));   // the pinned staging area is reused by the polls below
    }
    HIP_CHECK(hipMemsetAsync(ds.itemReserve.ptr, 0, sizeof(uint2) * ((slots + 63u) / 64u), stream));
    if (rp.maxDepth == 0) HIP_CHECK(hipMemsetAsync(ds.itemAccum.ptr, 0, sizeof(float4) * rp.itemCount, stream));

    const auto wall0 = std::chrono::steady_clock::now();
    launchGenerate(rp, pool, stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(ds.groupEvents[0], stream));
    for (uint32_t g = 1; g < groupCount; ++g) HIP_CHECK(hipStreamWaitEvent(groups[g].stream, ds.groupEvents[0], 0));
    uint64_t iterations = 0;
    // Worst case: every sample runs maxDepth bounces in sequence on its slot.
    // (a subsurface random walk adds up to sssMaxSteps iterations to a bounce)
    const uint64_t perBounce = ((rp.mediaMode & PTR_METAL_SSS) && rp.sssMode == 2u && ds.hasRandomWalkMaterial) ? 1ull + rp.sssMaxSteps : 1ull;
    const uint64_t maxIterations = static_cast<uint64_t>(rp.maxDepth) * perBounce * ((itemCount64 + slots - 1) / slots + 1) + 8;
    // Phase 1: while unclaimed work items remain nobody needs to count survivors; the host looks at the item head
    // only when it expects it to be nearly exhausted (items are claimed at a steady rate, so after the first look the
    // next one is scheduled at 3/4 of the predicted remaining iterations).  Phase 2 (queue dry): k_shade counts live
    // slots and every group is polled every 4 iterations until it has none left (k_extend reports how many live
    // slots it traced: one atomic per persistent wave).  A poll joins all streams, which
    // costs the overlap between groups once; polling every 4 iterations throughout was 5 % slower.  (Polling through
    // events without joining was tried: the host then runs up to a dozen empty iterations past the end - no gain.)
    constexpr uint64_t kPollEvery = 4;
    constexpr uint64_t kPollDry = 2;   // once the queue is dry: how often the live slots are counted (the hand-over to the tail kernels hangs on it)
    const ptr::Knobs knobs = ptr::readKnobs();
    const bool tracePolls = knobs.verbosePolls;   // debugging aid: live slots per poll
    uint64_t nextCheck = kPollEvery;
    bool queueDry = false;
    bool runTail = false;   // the last paths are handed to the tail kernels
    while (rp.maxDepth > 0) {
        const uint32_t ring = static_cast<uint32_t>(iterations % kAliveRing);
        for (Group& gr : groups) {
            if (gr.done) continue;
            uint32_t* aliveSlot = gr.scalars + kAliveBase + ring;
            gr.cfg.feederChunk = gr.feederChunk;
            // work heads and the next live-slot counter are cleared by k_shade (all zero at the start of the frame)
            const ShadeResets resets{gr.scalars + 1, gr.scalars + 2, gr.scalars + kAliveBase + (ring + 1u) % kAliveRing,
                                     (queueDry && gr.feederChunk > ds.feederChunk) ? 1u : 0u};
            {   // this iteration's counters, and the set k_shade clears for the next one
                gr.pool.connectCount = gr.connectCounts + connectCountWords * (iterations & 1u);
                gr.pool.connectClear = gr.connectCounts + connectCountWords * ((iterations + 1u) & 1u);
            }
            if (queueDry && gr.busyStage == 0u) gr.busyStage = 1u;   // the kernels decide per launch whether a list pays
            if (gr.busyStage != 0u) {
                // k_shade fills list `busyTurn` (counter set busyTurn % 3) and clears the set after it; in stage 2 this iteration's
                // k_extend and k_shade walk the list the previous iteration filled
                const uint32_t turn = gr.busyTurn;
                gr.pool.busyOut = gr.busyLists + connectListWords * (turn & 1u);
                gr.pool.busyCountOut = gr.busyCounts + connectCountWords * (turn % 3u);
                gr.pool.busyCountClear = gr.busyCounts + connectCountWords * ((turn + 1u) % 3u);
                if (gr.busyStage == 2u) {
                    gr.pool.busyIn = gr.busyLists + connectListWords * ((turn + 1u) & 1u);
                    gr.pool.busyCountIn = gr.busyCounts + connectCountWords * ((turn + 2u) % 3u);
                }
            }
            timedLaunch(0, gr.stream, [&] { launchExtend(ds.view, gr.pool, gr.cfg, queueDry ? aliveSlot : nullptr, count, gr.stream); });
            // k_shade's list instantiation claims leftover work items lane by lane and is a fifth slower than the plain kernel while
            // it still walks the slots: it is launched once the last count of live slots says its list is about to pay
            PathPool shadePool = gr.pool;
            if (!gr.shadeListed) shadePool.busyIn = nullptr;
            if (overlap && iterations > 0) HIP_CHECK(hipStreamWaitEvent(gr.stream, ds.sideEvents[2 * static_cast<uint32_t>(&gr - groups.data()) + 1], 0));
            timedLaunch(1, gr.stream, [&] { launchShade(rp, ds.view, shadePool, resets, count, gr.stream); });
            if (overlap) {
                // k_connect of this iteration on the side stream, after this k_shade; the next k_shade waits for it (above)
                const uint32_t gi = static_cast<uint32_t>(&gr - groups.data());
                hipStream_t side = ds.sideStreams[gi];
                HIP_CHECK(hipEventRecord(ds.sideEvents[2 * gi], gr.stream));          // shade(i) done
                HIP_CHECK(hipStreamWaitEvent(side, ds.sideEvents[2 * gi], 0));
                LaunchConfig ccfg = gr.cfg;
                ccfg.spill = ds.spill.ptr + (ds.maxPoolGroups() + gi) * spillWords;
                timedLaunch(2, side, [&] { launchConnect(rp, ds.view, gr.pool, ccfg, count, side); });
                HIP_CHECK(hipEventRecord(ds.sideEvents[2 * gi + 1], side));           // connect(i) done
            } else
            timedLaunch(2, gr.stream, [&] { launchConnect(rp, ds.view, gr.pool, gr.cfg, count, gr.stream); });
            if (gr.busyStage != 0u) {
                gr.busyStage = 2u;
                ++gr.busyTurn;
            }
        }
        ++iterations;
        if (iterations >= nextCheck || iterations >= maxIterations) {
            HIP_CHECK(hipGetLastError());   // a launch that failed (bad configuration, out of resources) must not pass for a slow frame
            bool headsCopied = false;
            for (uint32_t g = 0; g < groupCount; ++g) {
                Group& gr = groups[g];
                if (gr.done) continue;
                if (queueDry) {
                    HIP_CHECK(hipMemcpyAsync(ds.pinnedAlive + g, gr.scalars + kAliveBase + ring, sizeof(uint32_t), hipMemcpyDeviceToHost, gr.stream));
                } else if (!headsCopied) {
                    HIP_CHECK(hipMemcpy2DAsync(ds.pinnedAlive + kPinnedHeadsOffset, sizeof(uint32_t), pool.nextItem,
                                               sizeof(uint32_t) * kItemHeadStride, sizeof(uint32_t), kItemHeads, hipMemcpyDeviceToHost, gr.stream));
                    headsCopied = true;
                }
            }
            bool allDone = true;
            for (uint32_t g = 0; g < groupCount; ++g) {
                Group& gr = groups[g];
                if (gr.done) continue;
                HIP_CHECK(hipStreamSynchronize(gr.stream));
                if (queueDry) {
                    if (ds.pinnedAlive[g] == 0u) gr.done = true;
                    if (tracePolls) {
                        std::fprintf(stderr, "[poll] iteration %llu group %u live %u of %u chunk %u\n", static_cast<unsigned long long>(iterations), g,
                                     ds.pinnedAlive[g], gr.pool.slots, gr.feederChunk);
                    }
                    // the fewer live slots, the bigger the chunks the work list is claimed in (see WaveFeeder): a chunk
                    // should still hold about as many live slots as a full one does when the pool is full
                    // up to the point where the static first chunks of the resident waves cover the whole list and
                    // the head is not touched at all
                    const uint32_t thin = gr.pool.slots / std::max(ds.pinnedAlive[g], 1u);
                    const uint32_t waves = std::max(gr.cfg.traceGrid * (kTraceGridUnit / 64u), 1u);
                    const uint32_t perWave = ((gr.pool.slots + waves - 1u) / waves + 63u) / 64u * 64u;
                    const uint32_t cap = std::max(perWave, ds.feederChunk);
                    gr.feederChunk = std::min(cap, ds.feederChunk * std::max(thin, 1u));
                    gr.shadeListed = static_cast<uint64_t>(ds.pinnedAlive[g]) * 5u < static_cast<uint64_t>(gr.pool.slots) * 2u;   // < 40 % live
                }
                allDone = allDone && gr.done;
            }
            if (queueDry && !allDone && ds.tailBelow > 0) {
                uint64_t live = 0;
                for (uint32_t g = 0; g < groupCount; ++g) {
                    if (!groups[g].done) live += ds.pinnedAlive[g];
                }
                if (live <= ds.tailBelow) {
                    runTail = true;
                    break;
                }
            }
            uint64_t head = slots;   // items claimed so far = pre-assigned + what every range head has handed out
            if (!queueDry) {
                const uint32_t* heads = ds.pinnedAlive + kPinnedHeadsOffset;
                for (uint32_t k = 0; k < kItemHeads; ++k) {
                    const uint64_t lo = std::min<uint64_t>(static_cast<uint64_t>(rp.itemHeadFirst) + static_cast<uint64_t>(k) * rp.itemsPerHead, rp.itemCount);
                    const uint64_t hi = std::min<uint64_t>(lo + rp.itemsPerHead, rp.itemCount);
                    head += std::min<uint64_t>(std::max<uint64_t>(heads[k], lo), hi) - lo;
                }
            }
            if (allDone) break;
            nextCheck = iterations + (queueDry && ds.tailBelow > 0 ? kPollDry : kPollEvery);
            if (!queueDry) {
                if (head >= rp.itemCount) {
                    queueDry = true;
                } else if (head > slots) {
                    const double perIteration = static_cast<double>(head - slots) / static_cast<double>(iterations);
                    const double left = static_cast<double>(rp.itemCount - head) / std::max(perIteration, 1.0);
                    nextCheck = iterations + std::max<uint64_t>(kPollEvery, static_cast<uint64_t>(left * 0.75));
                }
            }
            if (iterations >= maxIterations) throw HipError{"wavefront loop did not terminate"};
        }
    }
    for (uint32_t g = 0; overlap && g < groupCount; ++g) HIP_CHECK(hipStreamWaitEvent(groups[g].stream, ds.sideEvents[2 * g + 1], 0));
    for (uint32_t g = 1; g < groupCount; ++g) {
        HIP_CHECK(hipEventRecord(ds.groupEvents[g], groups[g].stream));
        HIP_CHECK(hipStreamWaitEvent(stream, ds.groupEvents[g], 0));
    }
    if (runTail) {
        HIP_CHECK(hipMemsetAsync(ds.tailWords.ptr, 0, 4 * sizeof(uint32_t), stream));
        timedLaunch(3, stream, [&] { launchTail(rp, ds.view, pool, groups[0].cfg, ds.tailList.ptr, ds.tailWords.ptr, ds.tailWords.ptr + 1, count, stream); });
    }
    launchResolve(rp, pool, parts, dOut, stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->totalSeconds = seconds;
        stats->avgMsPerSample = seconds * 1000.0 / rp.spp;
        stats->uploadSeconds = ds.uploadSeconds;
        stats->samples = static_cast<uint64_t>(localPixels) * rp.spp;
        if (knobs.verboseLaunches && !spans.empty()) {   // debugging aid: when each launch ran (ms from the first)
            for (const Span& s : spans) {
                float t0 = 0.0f, t1 = 0.0f;
                HIP_CHECK(hipEventElapsedTime(&t0, spans.front().a, s.a));
                HIP_CHECK(hipEventElapsedTime(&t1, spans.front().a, s.b));
                std::fprintf(stderr, "[launch] kind %d  start %.3f  end %.3f  (%.3f ms)  stream %p\n", s.kind, t0, t1, t1 - t0, s.stream);
            }
        }
        for (const Span& s : spans) {
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
            if (s.kind == 0) {
                stats->traceKernelMs += ms;
                ++stats->traceLaunches;
            } else if (s.kind == 1) {
                stats->shadeKernelMs += ms;
            } else if (s.kind == 2) {
                stats->shadowKernelMs += ms;
            } else {
                stats->tailKernelMs += ms;
            }
        }
        if (count) {
            uint64_t c[kCounterSlots];
            HIP_CHECK(hipMemcpy(c, ds.counters.ptr, sizeof(c), hipMemcpyDeviceToHost));
            stats->primaryRays = c[kCntPrimaryRays];
            stats->extendRays = c[kCntExtendRays];
            stats->shadowRays = c[kCntShadowRays];
            stats->extendNodesVisited = c[kCntExtendNodes];
            stats->extendLeafPrimTests = c[kCntExtendPrims];
            stats->nodesVisited = c[kCntExtendNodes] + c[kCntShadowNodes];
            stats->leafPrimTests = c[kCntExtendPrims] + c[kCntShadowPrims];
            stats->shadedHits = c[kCntShadedHits];
            stats->triangleHits = c[kCntTriangleHits];
            stats->shadowEarlyExits = c[kCntShadowEarlyExit];
            if (knobs.verboseSteps) {   // lane utilisation of k_extend's step loop (counting build)
                const double nodeLanes = static_cast<double>(c[kCntExtendNodes] - c[kCntExtendLeaves]), primLanes = static_cast<double>(c[kCntExtendPrims]);
                const double nodeSlots = static_cast<double>(c[kCntExtendWaveNodeSteps]), primSlots = static_cast<double>(c[kCntExtendWavePrimSteps]);
                std::fprintf(stderr, "[steps] k_extend: %.3g rays; node steps %.3g lane / %.3g slots = %.3f; prim steps %.3g lane / %.3g slots = %.3f; "
                                     "refill passes %.3g (x64 lanes)\n",
                             static_cast<double>(c[kCntExtendRays]), nodeLanes, nodeSlots, nodeLanes / std::max(nodeSlots, 1.0), primLanes, primSlots,
                             primLanes / std::max(primSlots, 1.0), static_cast<double>(c[kCntExtendRefillPasses]));
                const double votes = static_cast<double>(c[kCntExtendVoteIterations]) / 64.0;
                std::fprintf(stderr, "[steps] k_extend: %.3g vote iterations; lanes holding a ray %.1f / 64 on average, of which at a leaf %.1f\n", votes,
                             static_cast<double>(c[kCntExtendActiveLanes]) / std::max(votes, 1.0),
                             static_cast<double>(c[kCntExtendLeafLanes]) / std::max(votes, 1.0));
                const double shadeLanes = std::max(static_cast<double>(c[kCntShadeWaves]), 1.0);
                std::fprintf(stderr, "[steps] k_shade: %.3g wave visits; share of their lanes at each stage: ray traced %.3f, surface hit %.3f, of which a light "
                                     "%.3f; light sample evaluated %.3f, tested against the light's own triangles %.3f, shadow ray queued %.3f; BSDF sampled "
                                     "%.3f; new work item wanted %.3f\n",
                             shadeLanes / 64.0, c[kCntShadeAlive] / shadeLanes, c[kCntShadeSurface] / shadeLanes, c[kCntShadeEmitter] / shadeLanes,
                             c[kCntShadeLightEval] / shadeLanes, c[kCntShadeLightPretest] / shadeLanes, c[kCntShadeLightStored] / shadeLanes,
                             c[kCntShadeBsdfSample] / shadeLanes, c[kCntShadeNeedItem] / shadeLanes);
                {
                    static const char* names[kShadeParts] = {"loads + landing", "background", "surface reconstruction", "emitter", "light sample", "env sample",
                                                             "BSDF sample + next ray", "work item + camera ray", "stores + lists", "subsurface walk"};
                    double waveTotal = 0.0;
                    for (uint32_t k = 0; k < kShadeParts; ++k) waveTotal += static_cast<double>(c[kCntShadeWaveTicks + k]);
                    std::fprintf(stderr, "[steps] k_shade parts (clock ticks between the part's first and last instruction, waits included): share of the waves' time | lanes busy\n");
                    for (uint32_t k = 0; k < kShadeParts; ++k) {
                        const double wave = static_cast<double>(c[kCntShadeWaveTicks + k]), lane = static_cast<double>(c[kCntShadeLaneTicks + k]);
                        if (wave <= 0.0) continue;
                        std::fprintf(stderr, "[steps]   %-24s %5.1f %% | %.3f\n", names[k], 100.0 * wave / std::max(waveTotal, 1.0), lane / (64.0 * wave));
                    }
                }
                std::fprintf(stderr, "[steps] k_extend: refill passes take %.1f %% of the waves' time in the kernel\n",
                             100.0 * static_cast<double>(c[kCntExtendRefillTicks]) / std::max(static_cast<double>(c[kCntExtendWaveTicks]), 1.0));
            }
        }
    }
}

// A frame.  One accumulator per sample has to fit in a quarter of the free device memory (at most 16 GiB); a frame with
// more samples than that is rendered in several passes of equal sample counts whose per-pixel sums add up in the output
// buffer.  (Folding C samples into one work item instead keeps one pass but lengthens the end-of-frame drain: 4096 spp of
// config 2 as items of 9 samples ran at 1116 Msamples/s.)  The sample streams do not depend on the split.
void renderBands(PtrDeviceScene& ds, const PtrSettings& settings, uint32_t spp, uint32_t part, uint32_t parts, float* dOut,
                 hipStream_t stream, int mode, PtrRenderStats* stats) {
    if (settings.width == 0 || settings.height == 0) throw HipError{"render size must be non-zero"};
    if (parts == 0 || part >= parts) throw HipError{"bad partition"};
    HIP_CHECK(hipSetDevice(ds.device));
    spp = std::max(1u, spp);
    uint32_t passes = 1u;
    {
        const uint64_t budget = itemBudgetBytes(ds);
        uint64_t maxItems = std::min<uint64_t>(budget / sizeof(float4), 0xFFFFFFF0ull);
        if (const uint64_t forced = ptr::readKnobs().maxItems) maxItems = forced;   // test knob
        // counted on the whole frame, not on this partition: every partition then splits the samples the same way and the image
        // stays bit-identical whatever the number of partitions
        const uint64_t pixels = static_cast<uint64_t>(settings.width) * settings.height;
        const uint64_t perPixel = std::max<uint64_t>(1u, maxItems / std::max<uint64_t>(pixels, 1u));   // samples per pixel that fit in one pass
        passes = static_cast<uint32_t>((spp + perPixel - 1u) / perPixel);
    }
    if (passes <= 1u) {
        renderPass(ds, settings, spp, 0u, spp, 3u, part, parts, dOut, stream, mode, stats);
        return;
    }
    const uint32_t perPass = (spp + passes - 1u) / passes;
    PtrRenderStats sum{};
    uint32_t done = 0u;
    for (uint32_t p = 0; done < spp; ++p) {
        const uint32_t n = std::min(perPass, spp - done);
        const uint32_t flags = (p == 0u ? 1u : 0u) | (done + n >= spp ? 2u : 0u);
        PtrRenderStats one{};
        renderPass(ds, settings, n, done, spp, flags, part, parts, dOut, stream, mode, stats ? &one : nullptr);
        if (stats) {
            sum.totalSeconds += one.totalSeconds;
            sum.traceKernelMs += one.traceKernelMs;
            sum.shadeKernelMs += one.shadeKernelMs;
            sum.shadowKernelMs += one.shadowKernelMs;
            sum.tailKernelMs += one.tailKernelMs;
            sum.traceLaunches += one.traceLaunches;
            sum.samples += one.samples;
            const double keepSeconds = sum.totalSeconds, keepTrace = sum.traceKernelMs, keepShade = sum.shadeKernelMs, keepShadow = sum.shadowKernelMs,
                         keepTail = sum.tailKernelMs;
            const uint64_t keepLaunches = sum.traceLaunches, keepSamples = sum.samples;
            sum = one;   // counters are cumulative on the device: the last pass reports the totals
            sum.totalSeconds = keepSeconds;
            sum.traceKernelMs = keepTrace;
            sum.shadeKernelMs = keepShade;
            sum.shadowKernelMs = keepShadow;
            sum.tailKernelMs = keepTail;
            sum.traceLaunches = keepLaunches;
            sum.samples = keepSamples;
        }
        done += n;
    }
    if (stats) {
        sum.avgMsPerSample = sum.totalSeconds * 1000.0 / spp;
        *stats = sum;
    }
}

}  // namespace

extern "C" {

int ptr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint32_t ptr_part_band_count(uint32_t height, uint32_t part_index, uint32_t part_count) {
    if (part_count == 0 || part_index >= part_count) return 0;
    const uint32_t bands = (height + PTR_BAND_ROWS - 1u) / PTR_BAND_ROWS;
    return bands > part_index ? (bands - part_index + part_count - 1u) / part_count : 0u;
}

int ptr_scene_upload(const PtrSceneDesc* scene, int device, PtrDeviceScene** out_scene, char* err, size_t err_cap) {
    if (!scene || !out_scene) {
        setErr(err, err_cap, "ptr_scene_upload: null argument");
        return 1;
    }
    if (ptr_device_count() <= device || device < 0) {
        setErr(err, err_cap, "ptr_scene_upload: no such HIP device (the HIP path has no CPU fallback)");
        return 2;
    }
    try {
        auto ds = std::make_unique<PtrDeviceScene>();
        ds->device = device;
        buildScene(*scene, *ds);
        *out_scene = ds.release();
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

void ptr_scene_release(PtrDeviceScene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int ptr_scene_info(const PtrDeviceScene* scene, uint64_t out[8]) {
    if (!scene || !out) return 1;
    std::memcpy(out, scene->info, sizeof(scene->info));
    return 0;
}

int ptr_scene_timings(const PtrDeviceScene* scene, double out[4]) {
    if (!scene || !out) return 1;
    std::memcpy(out, scene->timings, sizeof(scene->timings));
    return 0;
}

int ptr_scene_prepare_geometry(const PtrSceneDesc* scene, const char* cache_path, double* seconds, char* err, size_t err_cap) {
    if (!scene || !cache_path || !*cache_path) {
        setErr(err, err_cap, "ptr_scene_prepare_geometry: null argument");
        return 1;
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        ptr::PreparedGeometry pg;
        prepareGeometry(*scene, pg);
        std::string error;
        if (!ptr::WriteGeometryCache(cache_path, pg, ptr::SceneFingerprint(*scene), error)) throw HipError{error};
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_scene_upload_prepared(const PtrSceneDesc* scene, const char* cache_path, int device, PtrDeviceScene** out_scene, char* err, size_t err_cap) {
    if (!scene || !out_scene || !cache_path || !*cache_path) {
        setErr(err, err_cap, "ptr_scene_upload_prepared: null argument");
        return 1;
    }
    if (ptr_device_count() <= device || device < 0) {
        setErr(err, err_cap, "ptr_scene_upload_prepared: no such HIP device (the HIP path has no CPU fallback)");
        return 2;
    }
    try {
        auto ds = std::make_unique<PtrDeviceScene>();
        ds->device = device;
        buildScene(*scene, *ds, cache_path);
        *out_scene = ds.release();
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_render_bands_device(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t spp, uint32_t part_index,
                            uint32_t part_count, void* d_out_rgb, void* stream, int count_traversal,
                            PtrRenderStats* stats, char* err, size_t err_cap) {
    if (!scene || !settings || !d_out_rgb) {
        setErr(err, err_cap, "ptr_render_bands_device: null argument");
        return 1;
    }
    try {
        renderBands(*scene, *settings, spp, part_index, part_count, static_cast<float*>(d_out_rgb),
                    static_cast<hipStream_t>(stream), count_traversal, stats);
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_render_bands(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t spp, uint32_t part_index,
                     uint32_t part_count, float* out_rgb_bands, int count_traversal, PtrRenderStats* stats,
                     char* err, size_t err_cap) {
    if (!scene || !settings || !out_rgb_bands) {
        setErr(err, err_cap, "ptr_render_bands: null argument");
        return 1;
    }
    try {
        const size_t floats = static_cast<size_t>(ptr_part_band_count(settings->height, part_index, part_count)) * PTR_BAND_ROWS * settings->width * 3u;
        HIP_CHECK(hipSetDevice(scene->device));
        scene->outBands.ensure(floats);
        renderBands(*scene, *settings, spp, part_index, part_count, scene->outBands.ptr, nullptr, count_traversal, stats);
        HIP_CHECK(hipMemcpy(out_rgb_bands, scene->outBands.ptr, floats * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_render(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, int verbose, float* out_rgb,
               PtrRenderStats* stats, char* err, size_t err_cap) {
    if (!scene || !settings || !out_rgb) {
        setErr(err, err_cap, "ptr_render: null argument");
        return 1;
    }
    PtrDeviceScene* ds = nullptr;
    int rc = ptr_scene_upload(scene, 0, &ds, err, err_cap);
    if (rc != 0) return rc;
    const uint32_t bands = (settings->height + PTR_BAND_ROWS - 1u) / PTR_BAND_ROWS;
    std::vector<float> banded;
    try {
        banded.resize(static_cast<size_t>(bands) * PTR_BAND_ROWS * settings->width * 3u);
    } catch (const std::exception& e) {
        ptr_scene_release(ds);
        setErr(err, err_cap, std::string("exception: ") + e.what());
        return 1;
    }
    PtrRenderStats local{};
    rc = ptr_render_bands(ds, settings, spp, 0, 1, banded.data(), 0, &local, err, err_cap);
    if (rc == 0) {
        // with a single partition the band layout is the image itself (plus padding rows)
        std::memcpy(out_rgb, banded.data(), static_cast<size_t>(settings->width) * settings->height * 3u * sizeof(float));
        if (stats) *stats = local;
        if (verbose) {
            std::fprintf(stderr, "[ptr] BVH %llu nodes, %llu tris, %llu spheres; upload %.3f s; render %.3f s (trace %.1f ms, shade %.1f ms, connect %.1f ms)\n",
                         static_cast<unsigned long long>(ds->info[0]), static_cast<unsigned long long>(ds->info[2]),
                         static_cast<unsigned long long>(ds->info[3]), ds->uploadSeconds, local.totalSeconds,
                         local.traceKernelMs, local.shadeKernelMs, local.shadowKernelMs);
        }
    }
    ptr_scene_release(ds);
    return rc;
}

// A frame on several devices: the scene is prepared once on the host, every device gets its own copy (uploads run in
// parallel, one host thread per device), renders the bands b = rank (mod n) and hands them to the first device over the
// fabric (hipMemcpyPeerAsync: xGMI between the GPUs of a node), which interleaves them into the image.
// device_ids may name the same device more than once (how the single-GPU tests exercise the whole path).
static int renderMulti(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, const int* device_ids, int n, int verbose,
                       float* out_rgb, PtrRenderStats* stats, char* err, size_t err_cap) {
    try {
        if (settings->width == 0 || settings->height == 0) throw HipError{"render size must be non-zero"};
        const int available = ptr_device_count();
        if (available < 1) throw HipError{"no HIP device (the HIP path has no CPU fallback)"};
        // (test hook of ptr_debug_render_multi_on: an id given as -(id + 1) sends that partition's bands through the host-staging path)
        std::vector<int> devices(static_cast<size_t>(n));
        std::vector<char> forceStaged(static_cast<size_t>(n), 0);
        for (int i = 0; i < n; ++i) {
            forceStaged[static_cast<size_t>(i)] = device_ids[i] < 0 ? 1 : 0;
            devices[static_cast<size_t>(i)] = device_ids[i] < 0 ? -device_ids[i] - 1 : device_ids[i];
            if (devices[static_cast<size_t>(i)] >= available) throw HipError{"ptr_render_multi: no such HIP device"};
        }
        device_ids = devices.data();
        const auto t0 = std::chrono::steady_clock::now();
        PreparedScene prepared;
        prepareScene(*scene, prepared);

        const uint32_t parts = static_cast<uint32_t>(n);
        const uint32_t width = settings->width, height = settings->height;
        const size_t rowFloats = static_cast<size_t>(width) * 3u;
        std::vector<uint64_t> partOffset(parts + 1u, 0u);
        for (uint32_t p = 0; p < parts; ++p) {
            partOffset[p + 1u] = partOffset[p] + static_cast<uint64_t>(ptr_part_band_count(height, p, parts)) * PTR_BAND_ROWS * rowFloats;
        }
        const int rootDevice = device_ids[0];
        DeviceBuffer<float> gathered, image;
        DeviceBuffer<uint64_t> dOffsets;
        HIP_CHECK(hipSetDevice(rootDevice));
        gathered.ensure(partOffset[parts]);
        image.ensure(rowFloats * height);
        dOffsets.upload(partOffset.data(), parts);

        std::vector<std::unique_ptr<PtrDeviceScene>> scenes(parts);
        std::vector<PtrRenderStats> partStats(parts);
        std::vector<std::string> errors(parts);
        std::vector<double> uploadSeconds(parts, 0.0), renderSeconds(parts, 0.0);
        std::atomic<uint32_t> stagedParts{0};   // partitions whose bands went through host memory (no peer access)
        auto worker = [&](uint32_t p) {
            try {
                const auto w0 = std::chrono::steady_clock::now();
                auto ds = std::make_unique<PtrDeviceScene>();
                ds->device = device_ids[p];
                uploadScene(*scene, prepared, *ds);
                uploadSeconds[p] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
                const size_t floats = static_cast<size_t>(partOffset[p + 1u] - partOffset[p]);
                HIP_CHECK(hipSetDevice(ds->device));
                ds->outBands.ensure(floats);
                hipStream_t stream = nullptr;
                HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
                const auto r0 = std::chrono::steady_clock::now();
                renderBands(*ds, *settings, spp, p, parts, ds->outBands.ptr, stream, 0, &partStats[p]);
                // the partition's bands travel to the first device: a plain copy when it is local, device-to-device over the fabric
                // (xGMI between the GPUs of a node) when the two devices can address each other, through pinned host memory otherwise
                if (floats && ds->device == rootDevice && !forceStaged[p]) {
                    HIP_CHECK(hipMemcpyAsync(gathered.ptr + partOffset[p], ds->outBands.ptr, floats * sizeof(float), hipMemcpyDeviceToDevice, stream));
                } else if (floats) {
                    int direct = 0;
                    if (ds->device != rootDevice) HIP_CHECK(hipDeviceCanAccessPeer(&direct, ds->device, rootDevice));
                    if (forceStaged[p]) direct = 0;
                    if (direct) {
                        const hipError_t enabled = hipDeviceEnablePeerAccess(rootDevice, 0);   // (this thread's current device is ds->device)
                        if (enabled != hipSuccess && enabled != hipErrorPeerAccessAlreadyEnabled) direct = 0;
                        (void)hipGetLastError();
                    }
                    if (direct) {
                        HIP_CHECK(hipMemcpyPeerAsync(gathered.ptr + partOffset[p], rootDevice, ds->outBands.ptr, ds->device, floats * sizeof(float), stream));
                    } else {
                        std::fprintf(stderr, "[ptr] device %d does not address device %d directly: its bands go through pinned host memory\n", ds->device, rootDevice);
                        float* staging = nullptr;
                        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&staging), floats * sizeof(float), hipHostMallocDefault));
                        hipError_t copied = hipMemcpyAsync(staging, ds->outBands.ptr, floats * sizeof(float), hipMemcpyDeviceToHost, stream);
                        if (copied == hipSuccess) copied = hipStreamSynchronize(stream);
                        if (copied == hipSuccess) copied = hipSetDevice(rootDevice);
                        if (copied == hipSuccess) copied = hipMemcpy(gathered.ptr + partOffset[p], staging, floats * sizeof(float), hipMemcpyHostToDevice);
                        (void)hipSetDevice(ds->device);
                        (void)hipHostFree(staging);
                        HIP_CHECK(copied);
                        stagedParts.fetch_add(1);
                    }
                }
                HIP_CHECK(hipStreamSynchronize(stream));
                renderSeconds[p] = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
                HIP_CHECK(hipStreamDestroy(stream));
                scenes[p] = std::move(ds);
            } catch (const HipError& e) {
                errors[p] = e.message;
            } catch (const std::exception& e) {
                errors[p] = std::string("exception: ") + e.what();
            } catch (...) {
                errors[p] = "unknown exception";
            }
        };
        ptr::runOnThreads(parts, worker);
        for (uint32_t p = 0; p < parts; ++p) {
            if (!errors[p].empty()) throw HipError{"device " + std::to_string(device_ids[p]) + ": " + errors[p]};
        }
        HIP_CHECK(hipSetDevice(rootDevice));
        launchInterleaveBands(gathered.ptr, dOffsets.ptr, parts, width, height, image.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out_rgb, image.ptr, rowFloats * height * sizeof(float), hipMemcpyDeviceToHost));
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double slowestRender = 0.0, slowestUpload = 0.0;
        for (uint32_t p = 0; p < parts; ++p) {
            slowestRender = std::max(slowestRender, renderSeconds[p]);
            slowestUpload = std::max(slowestUpload, uploadSeconds[p]);
        }
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            // the integrate phase of the whole job: the slowest device's render + hand-over (devices run concurrently)
            stats->totalSeconds = slowestRender;
            stats->avgMsPerSample = slowestRender * 1000.0 / std::max(1u, spp);
            stats->uploadSeconds = prepared.seconds + slowestUpload;
            for (uint32_t p = 0; p < parts; ++p) {
                stats->samples += partStats[p].samples;
                stats->traceKernelMs += partStats[p].traceKernelMs;
                stats->shadeKernelMs += partStats[p].shadeKernelMs;
                stats->shadowKernelMs += partStats[p].shadowKernelMs;
                stats->tailKernelMs += partStats[p].tailKernelMs;
                stats->traceLaunches += partStats[p].traceLaunches;
            }
        }
        if (verbose) {
            std::fprintf(stderr, "[ptr] %d device(s): scene preparation %.3f s, slowest upload %.3f s, slowest render + hand-over %.3f s, whole call %.3f s\n", n,
                         prepared.seconds, slowestUpload, slowestRender, wall);
            for (uint32_t p = 0; p < parts; ++p) {
                std::fprintf(stderr, "[ptr]   device %d: %u bands, render %.3f s\n", device_ids[p], ptr_part_band_count(height, p, parts), renderSeconds[p]);
            }
        }
        for (uint32_t p = 0; p < parts; ++p) {
            if (scenes[p]) {
                (void)hipSetDevice(scenes[p]->device);
                scenes[p].reset();
            }
        }
        (void)hipSetDevice(rootDevice);
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_render_multi(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, int n_devices, int verbose, float* out_rgb,
                     PtrRenderStats* stats, char* err, size_t err_cap) {
    if (!scene || !settings || !out_rgb) {
        setErr(err, err_cap, "ptr_render_multi: null argument");
        return 1;
    }
    const int available = ptr_device_count();
    if (available < 1) {
        setErr(err, err_cap, "ptr_render_multi: no HIP device (the HIP path has no CPU fallback)");
        return 2;
    }
    int n = n_devices <= 0 ? available : n_devices;
    if (n > available) {
        setErr(err, err_cap, "ptr_render_multi: " + std::to_string(n) + " devices requested, " + std::to_string(available) + " visible");
        return 2;
    }
    // never more partitions than bands
    n = static_cast<int>(std::min<uint32_t>(static_cast<uint32_t>(n), std::max(1u, (settings->height + PTR_BAND_ROWS - 1u) / PTR_BAND_ROWS)));
    std::vector<int> ids(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) ids[static_cast<size_t>(i)] = i;
    return renderMulti(scene, settings, spp, ids.data(), n, verbose, out_rgb, stats, err, err_cap);
}

int ptr_debug_render_multi_on(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, const int* device_ids, int n, float* out_rgb,
                              PtrRenderStats* stats, char* err, size_t err_cap) {
    if (!scene || !settings || !out_rgb || !device_ids || n < 1) {
        setErr(err, err_cap, "ptr_debug_render_multi_on: bad argument");
        return 1;
    }
    return renderMulti(scene, settings, spp, device_ids, n, 0, out_rgb, stats, err, err_cap);
}

int ptr_trace_rays(PtrDeviceScene* scene, const float* rays, uint64_t n, int any_hit, PtrHit* out, PtrRenderStats* stats,
                   char* err, size_t err_cap) {
    if (!scene || (!rays && n) || (!out && n)) {
        setErr(err, err_cap, "ptr_trace_rays: null argument");
        return 1;
    }
    try {
        HIP_CHECK(hipSetDevice(scene->device));
        if (n == 0) return 0;
        scene->rayBatch.upload(reinterpret_cast<const float4*>(rays), n * 2);
        scene->hitBatch.ensure(n);
        HIP_CHECK(hipMemset(scene->counters.ptr, 0, sizeof(uint64_t) * kCounterSlots));
        LaunchConfig cfg{scene->traceGrid, scene->spill.ptr, scene->scalars.ptr + 1, scene->refillBelow};
        launchTraceRays(scene->view, scene->rayBatch.ptr, n, any_hit != 0, scene->hitBatch.ptr, cfg, scene->counters.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out, scene->hitBatch.ptr, n * sizeof(PtrHit), hipMemcpyDeviceToHost));
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            uint64_t c[kCounterSlots];
            HIP_CHECK(hipMemcpy(c, scene->counters.ptr, sizeof(c), hipMemcpyDeviceToHost));
            stats->nodesVisited = c[kCntExtendNodes] + c[kCntShadowNodes];
            stats->leafPrimTests = c[kCntExtendPrims] + c[kCntShadowPrims];
            stats->extendRays = any_hit ? 0 : n;
            stats->shadowRays = any_hit ? n : 0;
        }
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_render_aovs(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t sample_index, float* out_albedo, float* out_normal,
                    char* err, size_t err_cap) {
    if (!scene || !settings) {
        setErr(err, err_cap, "ptr_render_aovs: null argument");
        return 1;
    }
    try {
        if (settings->width == 0 || settings->height == 0) throw HipError{"render size must be non-zero"};
        HIP_CHECK(hipSetDevice(scene->device));
        RenderParams rp;
        fillRenderParams(*settings, 1u, rp);
        const size_t pixels = static_cast<size_t>(settings->width) * settings->height;
        DeviceBuffer<float4> albedo, normal;
        albedo.ensure(pixels);
        normal.ensure(pixels);
        LaunchConfig cfg{scene->traceGrid, scene->spill.ptr, scene->scalars.ptr + 1, scene->refillBelow};
        launchAovs(rp, scene->view, sample_index, albedo.ptr, normal.ptr, cfg, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        if (out_albedo) HIP_CHECK(hipMemcpy(out_albedo, albedo.ptr, pixels * sizeof(float4), hipMemcpyDeviceToHost));
        if (out_normal) HIP_CHECK(hipMemcpy(out_normal, normal.ptr, pixels * sizeof(float4), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

// ---- test-only entry points (include/ptr_debug.h) ----

int ptr_debug_render_signatures(PtrDeviceScene* scene, const PtrSettings* settings, float* out_rgb, uint32_t* out_signature, char* err,
                                size_t err_cap) {
    if (!scene || !settings || !out_signature) {
        setErr(err, err_cap, "ptr_debug_render_signatures: null argument");
        return 1;
    }
    try {
        const size_t pixels = static_cast<size_t>(settings->width) * settings->height;
        const size_t floats = static_cast<size_t>(ptr_part_band_count(settings->height, 0, 1)) * PTR_BAND_ROWS * settings->width * 3u;
        HIP_CHECK(hipSetDevice(scene->device));
        scene->outBands.ensure(floats);
        PtrRenderStats stats{};
        renderBands(*scene, *settings, 1u, 0u, 1u, scene->outBands.ptr, nullptr, 1, &stats);   // counting build, 1 spp: item = local pixel
        if (out_rgb) HIP_CHECK(hipMemcpy(out_rgb, scene->outBands.ptr, pixels * 3u * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<float4> items(pixels);
        std::vector<uint32_t> pixelOfLocal(pixels);
        HIP_CHECK(hipMemcpy(items.data(), scene->itemAccum.ptr, pixels * sizeof(float4), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(pixelOfLocal.data(), scene->pixelOfLocal.ptr, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t lp = 0; lp < pixels; ++lp) {
            uint32_t bits;
            std::memcpy(&bits, &items[lp].w, sizeof(bits));
            out_signature[pixelOfLocal[lp]] = bits;
        }
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_debug_texture_sample(PtrDeviceScene* scene, uint32_t texture, const float* in, uint64_t n, float* out, char* err, size_t err_cap) {
    if (!scene || (!in && n) || (!out && n)) {
        setErr(err, err_cap, "ptr_debug_texture_sample: null argument");
        return 1;
    }
    try {
        HIP_CHECK(hipSetDevice(scene->device));
        if (n == 0) return 0;
        DeviceBuffer<float> din;
        DeviceBuffer<float4> dout;
        din.upload(in, n * 3);
        dout.ensure(n);
        launchDebugTexSample(scene->view, texture, din.ptr, n, dout.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.ptr, n * sizeof(float4), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_debug_surface_hits(PtrDeviceScene* scene, const float* in, uint64_t n, float* out, char* err, size_t err_cap) {
    if (!scene || (!in && n) || (!out && n)) {
        setErr(err, err_cap, "ptr_debug_surface_hits: null argument");
        return 1;
    }
    try {
        HIP_CHECK(hipSetDevice(scene->device));
        if (n == 0) return 0;
        DeviceBuffer<float> din, dout;
        din.upload(in, n * 9);
        dout.ensure(n * 16);
        LaunchConfig cfg{scene->traceGrid, scene->spill.ptr, scene->scalars.ptr + 1, scene->refillBelow};
        launchDebugSurfaceHits(scene->view, din.ptr, n, dout.ptr, cfg, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.ptr, n * 16 * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_debug_exact_division(uint32_t d, const uint32_t* n, uint64_t count, uint32_t* out) {
    if (d == 0u || (!n && count) || (!out && count)) return 1;
    const DivU32 by = makeDivU32(d);
    for (uint64_t i = 0; i < count; ++i) out[i] = by.quotient(n[i]);
    return 0;
}

int ptr_debug_shade_kernel_set(const PtrDeviceScene* scene, const PtrSettings* settings, int count, uint32_t* out) {
    if (!scene || !settings || !out) return 1;
    RenderParams rp;
    fillRenderParams(*settings, 1, rp);
    *out = shadeKernelSet(rp, scene->view, count != 0);
    return 0;
}

int ptr_debug_eval_bsdf(const PtrMaterial* material, const PtrSettings* settings, const float* in, uint64_t n, float* out,
                        char* err, size_t err_cap) {
    try {
        if (ptr_device_count() < 1) throw HipError{"no HIP device (the HIP path has no CPU fallback)"};
        HIP_CHECK(hipSetDevice(0));
        std::vector<float> m;
        compactMaterial(*material, m);
        RenderParams rp;
        fillRenderParams(*settings, 1, rp);
        DeviceBuffer<float4> dm;
        DeviceBuffer<float> din, dout;
        dm.upload(reinterpret_cast<const float4*>(m.data()), kMaterialVec4);
        din.upload(in, n * 12);
        dout.ensure(n * 5);
        launchDebugEvalBsdf(dm.ptr, rp, din.ptr, n, dout.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.ptr, n * 5 * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_debug_sample_bsdf(const PtrMaterial* material, const PtrSettings* settings, const float* in, const uint32_t* front_face,
                          const uint32_t* rng_states, uint64_t n, float* out, uint32_t* out_states, char* err, size_t err_cap) {
    try {
        if (ptr_device_count() < 1) throw HipError{"no HIP device (the HIP path has no CPU fallback)"};
        HIP_CHECK(hipSetDevice(0));
        std::vector<float> m;
        compactMaterial(*material, m);
        RenderParams rp;
        fillRenderParams(*settings, 1, rp);
        DeviceBuffer<float4> dm;
        DeviceBuffer<float> din, dout;
        DeviceBuffer<uint32_t> dfront, drng, drngOut;
        dm.upload(reinterpret_cast<const float4*>(m.data()), kMaterialVec4);
        din.upload(in, n * 9);
        dfront.upload(front_face, n);
        drng.upload(rng_states, n);
        dout.ensure(n * 8);
        drngOut.ensure(n);
        launchDebugSampleBsdf(dm.ptr, rp, din.ptr, dfront.ptr, drng.ptr, n, dout.ptr, drngOut.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.ptr, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(out_states, drngOut.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

int ptr_debug_camera_rays(const PtrSettings* settings, const uint32_t* xys, uint64_t n, float* out, uint32_t* out_states,
                          char* err, size_t err_cap) {
    try {
        if (ptr_device_count() < 1) throw HipError{"no HIP device (the HIP path has no CPU fallback)"};
        HIP_CHECK(hipSetDevice(0));
        RenderParams rp;
        fillRenderParams(*settings, 1, rp);
        DeviceBuffer<uint32_t> dxy, drng;
        DeviceBuffer<float> dout;
        dxy.upload(xys, n * 3);
        dout.ensure(n * 6);
        drng.ensure(n);
        launchDebugCameraRays(rp, dxy.ptr, n, dout.ptr, drng.ptr, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.ptr, n * 6 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(out_states, drng.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return 0;
    }
    PTR_CATCH_ALL(err, err_cap)
}

}  // extern "C"
