// BVH traversal + primitive tests for gfx950 (closest-hit and any-hit).
//
// Replaces traverse_bvh_triangles_segment / trace_scene_software of the reference's Metal kernel
// (shaders/pathtrace.metal:1971-2165, 2266-2382) with Embree-path semantics (one world-space BVH over
// mesh triangles, rectangle halves and spheres; true closest hit; tnear = 1e-4):
//   triangle test  = Embree Moeller-Trumbore (external/embree/kernels/geometry/triangle_intersector_moeller.h:72-115)
//   sphere test    = Embree SPHERE_POINT   (external/embree/kernels/geometry/sphere_intersector.h:80-121)
//
// MI355X mapping: one lane = one ray; a 64 B node (both child boxes + child refs) is 4 dwordx4 loads per
// lane; leaves are encoded in the parent's child reference so a leaf costs no node fetch; the traversal
// stack lives in LDS as stack[level][lane] (lane-private column => no bank conflicts at any divergence),
// with levels >= kLdsStackLevels spilling to a lane-interleaved HBM area so occupancy is not bounded by
// worst-case tree depth.
#pragma once

#include "device_types.h"
#include "vec.h"

namespace ptrk {

struct TraceCounters {
    uint32_t nodes;   // internal nodes fetched + leaves visited (== "nodes popped" of a reference-layout BVH)
    uint32_t prims;   // primitive tests
    // counting build only, lane-utilisation bookkeeping: steps the WAVE executed (every lane counts them, so the wave sum
    // is 64 x the number of steps) and leaves this lane entered
    uint32_t waveNodeSteps = 0, wavePrimSteps = 0, leaves = 0;
};

// LDS words are addressed through an address-space-3 pointer so the stack always compiles to ds_read/ds_write
// (a generic pointer made the compiler merge the LDS and spill paths into one flat_load behind a branch).
typedef __attribute__((address_space(3))) uint32_t LdsWord;

struct LaneStack {
    LdsWord* lds;         // &ldsStack[threadIdx.x]; stride kTraceBlock
    uint32_t* spill;      // spill area of this launch (wave-uniform: stays in SGPRs); lane column = global thread id
    uint32_t spillStride;
    uint32_t limit;       // entries the stack can hold: kLdsStackLevels + the spill levels the host allocated for this scene's tree (SceneView::stackLimit)
    uint32_t sp;
    // the lane's column is recomputed on the (rare) spill path instead of keeping a 64-bit pointer alive per lane
    __device__ __forceinline__ uint32_t* spillSlot(uint32_t level) const {
        const uint32_t column = blockIdx.x * kTraceBlock + threadIdx.x;
        return spill + static_cast<size_t>(level - kLdsStackLevels) * spillStride + column;
    }
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < kLdsStackLevels) {
            lds[sp * kTraceBlock] = v;
        } else if (sp < limit) {
            *spillSlot(sp) = v;
        } else {
            return;  // cannot happen: the host sizes the spill area from the depth of the tree it built (hip_backend.cpp)
        }
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return (sp < kLdsStackLevels) ? lds[sp * kTraceBlock] : *spillSlot(sp);
    }
};

// Buffer descriptors of the two arrays every traversal step reads.  raw_buffer_load_b128 always issues one
// 16 B load per call (plain loads of the 32 B / 48 B records were split into 3-5 narrower ones and the
// reference words sunk behind the box tests as dependent loads), takes a 32-bit byte offset instead of 64-bit
// address arithmetic, and returns zeros instead of faulting if an index were ever out of range.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct SceneMem {
    __amdgpu_buffer_rsrc_t nodes;   // quantised or float nodes, whichever the scene uses
    __amdgpu_buffer_rsrc_t tris;
    __amdgpu_buffer_rsrc_t wide;    // four-wide nodes (NODES == 2 instantiations only)
};

__device__ __forceinline__ SceneMem sceneMem(const SceneView& sc) {
    SceneMem m;
    const void* nodes = sc.useQuantized ? static_cast<const void*>(sc.qnodes) : static_cast<const void*>(sc.nodes);
    m.nodes = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(nodes), 0, sc.nodeBytes, 0x00020000);
    m.tris = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(sc.tris), 0, sc.triBytes, 0x00020000);
    m.wide = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(sc.wnodes), 0, sc.wideBytes, 0x00020000);
    return m;
}

__device__ __forceinline__ uint4 load16u(__amdgpu_buffer_rsrc_t r, uint32_t byteOffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byteOffset, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load16f(__amdgpu_buffer_rsrc_t r, uint32_t byteOffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byteOffset, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// The hit record is distance + primitive only: the barycentrics a shader needs are recomputed from the same operands
// by triangleUv (bit-identical to what the traversal's own test produced), so they cost neither registers in the
// step loop nor 8 B per ray in the path-state stream.
struct TraceHit {
    float t;
    uint32_t prim;   // kHitMiss, or leaf-order index (| kHitSphereBit for spheres)
};

// Slab test of one child box in the form t = plane*inv - org*inv (one fma per plane; `oi` = org*inv is
// computed once per ray).  fminf/fmaxf drop the NaNs of axis-parallel rays (inf - inf), which only makes the
// test more permissive; boxes are padded at build time (float nodes: 1e-5 relative, quantised nodes: one whole
// cell) and the final comparison carries 4 ulp of slack, so a box is never culled when a primitive inside it
// passes the exact test.  Box tests only prune: their rounding never reaches the reported hit.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool slabTest(f3 lo, f3 hi, f3 oi, f3 inv, float tnear, float tfar, float& entry) {
    // both planes of an axis in ONE v_pk_fma_f32 (packed fp32, op_sel broadcasts inv / oi, the negation is a free modifier):
    // 3 instructions per box instead of 6, the same IEEE fma per lane half
    const v2f tx = __builtin_elementwise_fma((v2f){lo.x, hi.x}, (v2f){inv.x, inv.x}, (v2f){-oi.x, -oi.x});
    const v2f ty = __builtin_elementwise_fma((v2f){lo.y, hi.y}, (v2f){inv.y, inv.y}, (v2f){-oi.y, -oi.y});
    const v2f tz = __builtin_elementwise_fma((v2f){lo.z, hi.z}, (v2f){inv.z, inv.z}, (v2f){-oi.z, -oi.z});
    const float ax = tx.x, bx = tx.y, ay = ty.x, by = ty.y, az = tz.x, bz = tz.y;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tnear));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tfar));
    entry = t0;
    return t0 <= t1 * 1.0000004f;
}

// The same test with the planes already in ray order (near = the plane the ray meets first on that axis, by the sign of its
// direction): for an ordered box min(a, b) / max(a, b) of slabTest ARE the near / far values (the fma is monotonic), so the result is
// the same bits - without the six min / max per box - and an inverted box (an unused place of a wide node) fails it on every axis.
__device__ __forceinline__ bool slabTestOrdered(f3 nearQ, f3 farQ, f3 oi, f3 inv, float tnear, float tfar, float& entry) {
    const v2f tx = __builtin_elementwise_fma((v2f){nearQ.x, farQ.x}, (v2f){inv.x, inv.x}, (v2f){-oi.x, -oi.x});
    const v2f ty = __builtin_elementwise_fma((v2f){nearQ.y, farQ.y}, (v2f){inv.y, inv.y}, (v2f){-oi.y, -oi.y});
    const v2f tz = __builtin_elementwise_fma((v2f){nearQ.z, farQ.z}, (v2f){inv.z, inv.z}, (v2f){-oi.z, -oi.z});
    const float t0 = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tnear));
    const float t1 = fminf(fminf(tx.y, ty.y), fminf(tz.y, tfar));
    entry = t0;
    return t0 <= t1 * 1.0000004f;
}

__device__ __forceinline__ bool triangleTest(f3 v0, f3 e1, f3 e2, f3 org, f3 dir, float tnear, float tfar,
                                             float& t, float& u, float& v) {
    const f3 Ng = cross(e2, e1);
    const f3 C = v0 - org;
    const f3 R = cross(C, dir);
    const float den = dot(Ng, dir);
    const float absDen = fabsf(den);
    const float sgn = signbit(den) ? -1.0f : 1.0f;
    const float U = dot(R, e2) * sgn;
    const float V = dot(R, e1) * sgn;
    if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && (U + V) <= absDen)) return false;
    const float T = dot(Ng, C) * sgn;
    if (!(absDen * tnear < T && T <= absDen * tfar)) return false;
    const float rcpAbsDen = 1.0f / absDen;
    t = T * rcpAbsDen;
    u = U * rcpAbsDen;
    v = V * rcpAbsDen;
    return true;
}

// Barycentrics of a triangle hit: the same expressions, in the same order, as triangleTest above.
__device__ __forceinline__ void triangleUv(f3 v0, f3 e1, f3 e2, f3 org, f3 dir, float& u, float& v) {
    const f3 Ng = cross(e2, e1);
    const f3 C = v0 - org;
    const f3 R = cross(C, dir);
    const float den = dot(Ng, dir);
    const float absDen = fabsf(den);
    const float sgn = signbit(den) ? -1.0f : 1.0f;
    const float U = dot(R, e2) * sgn;
    const float V = dot(R, e1) * sgn;
    const float rcpAbsDen = 1.0f / absDen;
    u = U * rcpAbsDen;
    v = V * rcpAbsDen;
}

__device__ __forceinline__ bool sphereTest(float4 s, f3 org, f3 dir, float tnear, float tfar, float& t) {
    const float rd2 = 1.0f / dot(dir, dir);
    const f3 c0 = mk3(s) - org;
    const float projC0 = dot(c0, dir) * rd2;
    const f3 perp = c0 - projC0 * dir;
    const float l2 = dot(perp, perp);
    const float r2 = s.w * s.w;
    if (!(l2 <= r2)) return false;
    const float td = sqrtf((r2 - l2) * rd2);
    const float tFront = projC0 - td;
    const float tBack = projC0 + td;
    const bool validFront = (tnear <= tFront) && (tFront <= tfar);
    const bool validBack = (tnear <= tBack) && (tBack <= tfar);
    if (!validFront && !validBack) return false;
    t = validFront ? tFront : tBack;
    return true;
}

// ---- stepwise traversal (persistent waves with lane refill) -------------------------------------------
// A lane owns one ray at a time and is either at an internal node or inside a leaf.  Each wave iteration runs
// ONE kind of step, chosen by majority vote over the lanes (travVote): a node step (one 64 B node, two slab
// tests) or a primitive step (one 48 B triangle / one sphere).  The branch is wave-uniform, so the wave never
// executes both bodies in one iteration, and lanes of the minority kind simply wait until they are the
// majority.  With the plain per-ray loop VALU lane utilisation was ~11 %; lane refill alone gave ~30 %.
struct Trav {
    f3 org, dir, inv;    // inv = 1/dir, or cell/dir when the scene uses quantised nodes
    f3 oi;               // org*inv in the space of the node boxes (world, or grid cells for quantised nodes)
    uint32_t selX, selY, selZ;   // four-wide nodes: v_perm selectors that put a child box's planes in ray order (travWideStep)
    float tnear;
    TraceHit hit;
    uint32_t cur;        // internal node index, or leaf reference while primitives remain
    uint32_t leafPos;    // next primitive inside the current leaf
    bool anyHit;
};

// NODES: 1 = 32 B quantised nodes, 0 = 64 B float nodes (compile-time choice of the persistent kernels), -1 = decided by the scene's
// flag at run time (the cold kernels: ray-batch queries, feature buffers, chains, the end-of-frame kernel)
template <int NODES = -1>
__device__ __forceinline__ bool travBegin(const SceneView& sc, Trav& t, f3 org, f3 dir, float tnear, float tfar, bool anyHit,
                                          LaneStack& stack) {
    t.org = org;
    t.dir = dir;
    // 1/dir clamped to a finite range: with the fma slab form an infinite reciprocal (a direction component of
    // exactly 0, e.g. a cosine sample with u1 = 0 -> the normal itself; ~1 ray in 2^24) turns both planes of that
    // axis into inf - inf = NaN, the axis stops culling, and that one ray walks most of the tree while its
    // persistent wave holds the whole launch (measured: k_extend 10x slower).  A clamped reciprocal treats the
    // component as +-1e-28: still "parallel" for any scene, and all box arithmetic stays finite.
    constexpr float kInvMax = 1.0e28f;
    // v_rcp_f32 (1 ulp) instead of IEEE divisions: these values only feed the box tests, which carry 4 ulp of slack
    // and padded boxes; the reported hit comes from the exact primitive tests on t.org / t.dir
    t.inv = mk3(fminf(fmaxf(__builtin_amdgcn_rcpf(dir.x), -kInvMax), kInvMax), fminf(fmaxf(__builtin_amdgcn_rcpf(dir.y), -kInvMax), kInvMax),
                fminf(fmaxf(__builtin_amdgcn_rcpf(dir.z), -kInvMax), kInvMax));
    if (NODES >= 1 || (NODES < 0 && sc.useQuantized)) {
        const f3 cell = mk3(sc.gridCell[0], sc.gridCell[1], sc.gridCell[2]);
        const f3 invCell = mk3(sc.gridInvCell[0], sc.gridInvCell[1], sc.gridInvCell[2]);
        const f3 orgQ = (org - mk3(sc.gridOrigin[0], sc.gridOrigin[1], sc.gridOrigin[2])) * invCell;
        t.inv = t.inv * cell;
        t.oi = orgQ * t.inv;
    } else {
        t.oi = org * t.inv;
    }
    // the child record is {lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, reference}: one v_perm_b32 per axis picks the
    // (near | far << 16) pair out of two of its words - which plane is near depends on the sign of the direction alone
    t.selX = t.inv.x < 0.0f ? 0x01000706u : 0x07060100u;   // perm(c.y, c.x): lo.x = bytes 0,1 of c.x; hi.x = bytes 2,3 of c.y
    t.selY = t.inv.y < 0.0f ? 0x03020504u : 0x05040302u;   // perm(c.z, c.x): lo.y = bytes 2,3 of c.x; hi.y = bytes 0,1 of c.z
    t.selZ = t.inv.z < 0.0f ? 0x01000706u : 0x07060100u;   // perm(c.z, c.y): lo.z = bytes 0,1 of c.y; hi.z = bytes 2,3 of c.z
    t.tnear = tnear;
    t.hit.t = tfar;
    t.hit.prim = kHitMiss;
    t.anyHit = anyHit;
    t.cur = sc.rootRef;
    t.leafPos = 0u;
    stack.sp = 0;
    if (sc.oversizeRef != kRefEmpty) {
        // the few triangles kept out of the tree (bvh_builder.cpp) come first: their hits shorten the ray before the walk
        if (sc.rootRef != kRefEmpty) stack.push(sc.rootRef);
        t.cur = sc.oversizeRef;
        return true;
    }
    return sc.rootRef != kRefEmpty;
}

__device__ __forceinline__ bool travAtLeaf(const Trav& t) { return (t.cur & kRefLeafBit) != 0u; }
// pops the next subtree; false when the traversal is complete
__device__ __forceinline__ bool travPop(Trav& t, LaneStack& stack) {
    if (stack.sp == 0) return false;
    t.cur = stack.pop();
    t.leafPos = 0u;
    return true;
}

__device__ __forceinline__ f3 gridLo(uint32_t w0, uint32_t w1) {
    return mk3(static_cast<float>(w0 & 0xFFFFu), static_cast<float>(w0 >> 16), static_cast<float>(w1 & 0xFFFFu));
}
__device__ __forceinline__ f3 gridHi(uint32_t w1, uint32_t w2) {
    return mk3(static_cast<float>(w1 >> 16), static_cast<float>(w2 & 0xFFFFu), static_cast<float>(w2 >> 16));
}

// Node step over a four-wide node (NODES == 2, SceneView::wnodes): four boxes, the hit children in order of entry distance (a
// five-exchange sorting network on (distance, reference) pairs), the nearest is walked next, the others are pushed far to near.
template <bool COUNT>
__device__ __forceinline__ bool travWideStep(const SceneMem& mem, Trav& t, LaneStack& stack, TraceCounters& cnt) {
    const uint32_t at = t.cur * 64u;
    const uint4 c0 = load16u(mem.wide, at), c1 = load16u(mem.wide, at + 16u), c2 = load16u(mem.wide, at + 32u), c3 = load16u(mem.wide, at + 48u);
    float k0, k1, k2, k3;
    // (an unused place holds an inverted box, bvh_builder.cpp BuildWideNodes: it fails the ordered test, its reference is not looked at)
    auto boxTest = [&](const uint4& c, float& k) {
        const uint32_t wx = __builtin_amdgcn_perm(c.y, c.x, t.selX), wy = __builtin_amdgcn_perm(c.z, c.x, t.selY), wz = __builtin_amdgcn_perm(c.z, c.y, t.selZ);
        const f3 nearQ = mk3(static_cast<float>(wx & 0xFFFFu), static_cast<float>(wy & 0xFFFFu), static_cast<float>(wz & 0xFFFFu));
        const f3 farQ = mk3(static_cast<float>(wx >> 16), static_cast<float>(wy >> 16), static_cast<float>(wz >> 16));
        return slabTestOrdered(nearQ, farQ, t.oi, t.inv, t.tnear, t.hit.t, k);
    };
    const bool h0 = boxTest(c0, k0), h1 = boxTest(c1, k1), h2 = boxTest(c2, k2), h3 = boxTest(c3, k3);
    k0 = h0 ? k0 : INFINITY;
    k1 = h1 ? k1 : INFINITY;
    k2 = h2 ? k2 : INFINITY;
    k3 = h3 ? k3 : INFINITY;
    uint32_t r0 = c0.w, r1 = c1.w, r2 = c2.w, r3 = c3.w;
    if (COUNT) ++cnt.nodes;
    t.leafPos = 0u;
    auto exchange = [](float& ka, uint32_t& ra, float& kb, uint32_t& rb) {
        const bool swap = kb < ka;
        const float kLo = swap ? kb : ka, kHi = swap ? ka : kb;
        const uint32_t rLo = swap ? rb : ra, rHi = swap ? ra : rb;
        ka = kLo;
        kb = kHi;
        ra = rLo;
        rb = rHi;
    };
    exchange(k0, r0, k1, r1);
    exchange(k2, r2, k3, r3);
    exchange(k0, r0, k2, r2);
    exchange(k1, r1, k3, r3);
    exchange(k1, r1, k2, r2);
    if (k3 < INFINITY) stack.push(r3);
    if (k2 < INFINITY) stack.push(r2);
    if (k1 < INFINITY) stack.push(r1);
    if (k0 < INFINITY) {
        t.cur = r0;
        return true;
    }
    return travPop(t, stack);
}

// Node step (t.cur is an internal node).  Returns false once the ray is finished.
template <bool COUNT, int NODES = -1>
__device__ __forceinline__ bool travNodeStep(const SceneView& sc, const SceneMem& mem, Trav& t, LaneStack& stack, TraceCounters& cnt) {
    if (NODES == 2) return travWideStep<COUNT>(mem, t, stack, cnt);
    uint32_t ref0, ref1;
    float e0, e1;
    bool h0, h1;
    // both halves of the node are fetched up front and both boxes tested without branching: a short-circuit
    // on the child reference made the compiler issue the second half as a dependent load
    if (NODES == 1 || (NODES < 0 && sc.useQuantized)) {
        const uint4 q0 = load16u(mem.nodes, t.cur * 32u), q1 = load16u(mem.nodes, t.cur * 32u + 16u);
        ref0 = q0.w;
        ref1 = q1.w;
        h0 = slabTest(gridLo(q0.x, q0.y), gridHi(q0.y, q0.z), t.oi, t.inv, t.tnear, t.hit.t, e0);
        h1 = slabTest(gridLo(q1.x, q1.y), gridHi(q1.y, q1.z), t.oi, t.inv, t.tnear, t.hit.t, e1);
    } else {
        const uint32_t at = t.cur * 64u;
        const float4 n0 = load16f(mem.nodes, at), n1 = load16f(mem.nodes, at + 16u), n2 = load16f(mem.nodes, at + 32u),
                     n3 = load16f(mem.nodes, at + 48u);
        ref0 = __float_as_uint(n0.w);
        ref1 = __float_as_uint(n1.w);
        h0 = slabTest(mk3(n0), mk3(n1), t.oi, t.inv, t.tnear, t.hit.t, e0);
        h1 = slabTest(mk3(n2), mk3(n3), t.oi, t.inv, t.tnear, t.hit.t, e1);
    }
    h0 = h0 & (ref0 != kRefEmpty);
    h1 = h1 & (ref1 != kRefEmpty);
    if (COUNT) ++cnt.nodes;
    const bool firstIs0 = e0 <= e1;
    const uint32_t nearRef = (h0 & (firstIs0 | !h1)) ? ref0 : ref1;
    const uint32_t farRef = firstIs0 ? ref1 : ref0;
    t.leafPos = 0u;
    if (h0 & h1) stack.push(farRef);
    if (h0 | h1) {
        t.cur = nearRef;
        return true;
    }
    return travPop(t, stack);
}

// Primitive step (t.cur is a leaf): tests primitive number t.leafPos of the leaf.  Returns false once finished.
template <bool COUNT>
__device__ __forceinline__ bool travPrimStep(const SceneView& sc, const SceneMem& mem, Trav& t, LaneStack& stack, TraceCounters& cnt) {
    const uint32_t cur = t.cur;
    const uint32_t first = cur & kRefOffsetMask;
    const uint32_t count = ((cur >> kRefCountShift) & 0xFu) + 1u;
    const uint32_t index = first + t.leafPos;
    if (COUNT) { ++cnt.prims; if (t.leafPos == 0u) { ++cnt.nodes; ++cnt.leaves; } }
    if (cur & kRefSphereBit) {
        float tt;
        if (sphereTest(sc.spheres[index], t.org, t.dir, t.tnear, t.hit.t, tt)) {
            t.hit.t = tt;
            t.hit.prim = index | kHitSphereBit;
            if (t.anyHit) return false;
        }
    } else {
        const uint32_t at = index * 48u;
        const float4 a = load16f(mem.tris, at), b = load16f(mem.tris, at + 16u), c = load16f(mem.tris, at + 32u);
        float tt, u, v;
        // (an any-hit query starts with the word it must not stop at in its hit word: kHitMiss, which no triangle carries, or the
        // meta word of the rectangle whose own two triangles a light connection ignores - wavefront.hip, kind-3 records)
        if (triangleTest(mk3(a), mk3(b), mk3(c), t.org, t.dir, t.tnear, t.hit.t, tt, u, v) && !(t.anyHit && __float_as_uint(b.w) == t.hit.prim)) {
            t.hit.t = tt;
            t.hit.prim = index;
            if (t.anyHit) return false;
        }
    }
    if (++t.leafPos < count) return true;
    return travPop(t, stack);
}

#ifndef PTR_EXTRA_NODE_STEPS
#define PTR_EXTRA_NODE_STEPS 3
#endif
#ifndef PTR_EXTRA_PRIM_STEPS
#define PTR_EXTRA_PRIM_STEPS 1
#endif
#ifndef PTR_PRIM_BIAS     // a primitive step is taken once the lanes at a leaf exceed 1/PTR_PRIM_BIAS of the lanes at a node
#define PTR_PRIM_BIAS 2
#endif
#ifndef PTR_REPEAT_NUM   // a step is repeated while at least NUM/DEN of the lanes that voted for it want it again
#define PTR_REPEAT_NUM 1
#define PTR_REPEAT_DEN 2
#endif

// One wave iteration for all traversing lanes: majority vote between node steps and primitive steps.
// Returns (per lane) false when that lane's ray has just finished.  Lanes not voted for return true unchanged.
template <bool COUNT, int NODES = -1>
__device__ __forceinline__ bool travVote(const SceneView& sc, const SceneMem& mem, Trav& t, bool active, LaneStack& stack,
                                         TraceCounters& cnt) {
    const bool wantsPrim = active && travAtLeaf(t);
    const bool wantsNode = active && !travAtLeaf(t);
    const int nPrim = __popcll(__ballot(wantsPrim));
    const int nNode = __popcll(__ballot(wantsNode));
    bool more = true;
    if (nNode >= nPrim * PTR_PRIM_BIAS) {
        if (COUNT) ++cnt.waveNodeSteps;
        if (wantsNode) more = travNodeStep<COUNT, NODES>(sc, mem, t, stack, cnt);
        // further node steps without another vote while most of these lanes land on an internal node again (the vote -
        // two ballots, two popcounts, the branch - costs about a fifth of a step)
#pragma unroll
        for (int extra = 0; extra < PTR_EXTRA_NODE_STEPS; ++extra) {
            const bool again = wantsNode && more && !travAtLeaf(t);
            if (static_cast<int>(__popcll(__ballot(again))) * PTR_REPEAT_DEN < nNode * PTR_REPEAT_NUM) break;
            if (COUNT) ++cnt.waveNodeSteps;
            if (again) more = travNodeStep<COUNT, NODES>(sc, mem, t, stack, cnt);
        }
    } else {
        if (COUNT) ++cnt.wavePrimSteps;
        if (wantsPrim) more = travPrimStep<COUNT>(sc, mem, t, stack, cnt);
#pragma unroll
        for (int extra = 0; extra < PTR_EXTRA_PRIM_STEPS; ++extra) {
            const bool again = wantsPrim && more && travAtLeaf(t);
            if (static_cast<int>(__popcll(__ballot(again))) * PTR_REPEAT_DEN < nPrim * PTR_REPEAT_NUM) break;
            if (COUNT) ++cnt.wavePrimSteps;
            if (again) more = travPrimStep<COUNT>(sc, mem, t, stack, cnt);
        }
    }
    return more;
}

// Whole-ray loop for one lane (ray-batch queries, MNEE chains): closest hit (ANY = false) or first hit
// (ANY = true).  Returns hit.prim == kHitMiss on a miss.  Same step functions as the persistent kernels.
template <bool ANY, bool COUNT>
__device__ __forceinline__ TraceHit traverse(const SceneView& sc, f3 org, f3 dir, float tnear, float tfar,
                                             LaneStack& stack, TraceCounters& cnt, uint32_t ignoreWord = kHitMiss) {
    const SceneMem mem = sceneMem(sc);
    Trav t;
    if (!travBegin(sc, t, org, dir, tnear, tfar, ANY, stack)) return t.hit;
    if (ANY) t.hit.prim = ignoreWord;
    bool more = true;
    while (more) {
        more = travAtLeaf(t) ? travPrimStep<COUNT>(sc, mem, t, stack, cnt) : travNodeStep<COUNT>(sc, mem, t, stack, cnt);
    }
    return t.hit;
}

}  // namespace ptrk
