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
};

struct LaneStack {
    uint32_t* lds;        // &ldsStack[threadIdx.x]; stride kTraceBlock
    uint32_t* spill;      // &spill[globalThread]; stride spillStride
    uint32_t spillStride;
    uint32_t sp;
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < kLdsStackLevels) {
            lds[sp * kTraceBlock] = v;
        } else if (sp < kTraversalStackDepth) {
            spill[static_cast<size_t>(sp - kLdsStackLevels) * spillStride] = v;
        } else {
            return;  // cannot happen: the builder bounds tree depth below kTraversalStackDepth
        }
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return (sp < kLdsStackLevels) ? lds[sp * kTraceBlock]
                                      : spill[static_cast<size_t>(sp - kLdsStackLevels) * spillStride];
    }
};

struct TraceHit {
    float t, u, v;
    uint32_t prim;   // kHitMiss, or leaf-order index (| kHitSphereBit for spheres)
};

// Slab test of one child box.  fminf/fmaxf drop NaNs from 0*inf, boxes are padded at build time, and the
// final comparison carries 4 ulp of slack, so a box is never culled when a primitive inside it passes
// the exact test.
__device__ __forceinline__ bool slabTest(f3 lo, f3 hi, f3 org, f3 inv, float tnear, float tfar, float& entry) {
    const float ax = (lo.x - org.x) * inv.x, bx = (hi.x - org.x) * inv.x;
    const float ay = (lo.y - org.y) * inv.y, by = (hi.y - org.y) * inv.y;
    const float az = (lo.z - org.z) * inv.z, bz = (hi.z - org.z) * inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tnear));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tfar));
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
    f3 orgQ;             // (org - gridOrigin) / cell  (quantised nodes only)
    float tnear;
    TraceHit hit;
    uint32_t cur;        // internal node index, or leaf reference while primitives remain
    uint32_t leafPos;    // next primitive inside the current leaf
    bool anyHit;
};

__device__ __forceinline__ bool travBegin(const SceneView& sc, Trav& t, f3 org, f3 dir, float tnear, float tfar, bool anyHit,
                                          LaneStack& stack) {
    t.org = org;
    t.dir = dir;
    t.inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    t.orgQ = mk3(0.0f);
    if (sc.useQuantized) {
        const f3 cell = mk3(sc.gridCell[0], sc.gridCell[1], sc.gridCell[2]);
        t.orgQ = (org - mk3(sc.gridOrigin[0], sc.gridOrigin[1], sc.gridOrigin[2])) / cell;
        t.inv = t.inv * cell;
    }
    t.tnear = tnear;
    t.hit.t = tfar;
    t.hit.u = 0.0f;
    t.hit.v = 0.0f;
    t.hit.prim = kHitMiss;
    t.anyHit = anyHit;
    t.cur = sc.rootRef;
    t.leafPos = 0u;
    stack.sp = 0;
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

// Node step (t.cur is an internal node).  Returns false once the ray is finished.
template <bool COUNT>
__device__ __forceinline__ bool travNodeStep(const SceneView& sc, Trav& t, LaneStack& stack, TraceCounters& cnt) {
    uint32_t ref0, ref1;
    float e0, e1;
    bool h0, h1;
    if (sc.useQuantized) {
        const uint4* n = sc.qnodes + static_cast<size_t>(t.cur) * 2u;
        const uint4 q0 = n[0], q1 = n[1];
        ref0 = q0.w;
        ref1 = q1.w;
        h0 = (ref0 != kRefEmpty) && slabTest(gridLo(q0.x, q0.y), gridHi(q0.y, q0.z), t.orgQ, t.inv, t.tnear, t.hit.t, e0);
        h1 = (ref1 != kRefEmpty) && slabTest(gridLo(q1.x, q1.y), gridHi(q1.y, q1.z), t.orgQ, t.inv, t.tnear, t.hit.t, e1);
    } else {
        const float4* n = sc.nodes + static_cast<size_t>(t.cur) * 4u;
        const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
        ref0 = __float_as_uint(n0.w);
        ref1 = __float_as_uint(n1.w);
        h0 = (ref0 != kRefEmpty) && slabTest(mk3(n0), mk3(n1), t.org, t.inv, t.tnear, t.hit.t, e0);
        h1 = (ref1 != kRefEmpty) && slabTest(mk3(n2), mk3(n3), t.org, t.inv, t.tnear, t.hit.t, e1);
    }
    if (COUNT) ++cnt.nodes;
    t.leafPos = 0u;
    if (h0 && h1) {
        const bool firstIs0 = e0 <= e1;
        stack.push(firstIs0 ? ref1 : ref0);
        t.cur = firstIs0 ? ref0 : ref1;
        return true;
    }
    if (h0) { t.cur = ref0; return true; }
    if (h1) { t.cur = ref1; return true; }
    return travPop(t, stack);
}

// Primitive step (t.cur is a leaf): tests primitive number t.leafPos of the leaf.  Returns false once finished.
template <bool COUNT>
__device__ __forceinline__ bool travPrimStep(const SceneView& sc, Trav& t, LaneStack& stack, TraceCounters& cnt) {
    const uint32_t cur = t.cur;
    const uint32_t first = cur & kRefOffsetMask;
    const uint32_t count = ((cur >> kRefCountShift) & 0xFu) + 1u;
    const uint32_t index = first + t.leafPos;
    if (COUNT) { ++cnt.prims; if (t.leafPos == 0u) ++cnt.nodes; }
    if (cur & kRefSphereBit) {
        float tt;
        if (sphereTest(sc.spheres[index], t.org, t.dir, t.tnear, t.hit.t, tt)) {
            t.hit.t = tt;
            t.hit.u = 0.0f;
            t.hit.v = 0.0f;
            t.hit.prim = index | kHitSphereBit;
            if (t.anyHit) return false;
        }
    } else {
        const float4* tp = sc.tris + static_cast<size_t>(index) * 3u;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        float tt, u, v;
        if (triangleTest(mk3(a), mk3(b), mk3(c), t.org, t.dir, t.tnear, t.hit.t, tt, u, v)) {
            t.hit.t = tt;
            t.hit.u = u;
            t.hit.v = v;
            t.hit.prim = index;
            if (t.anyHit) return false;
        }
    }
    if (++t.leafPos < count) return true;
    return travPop(t, stack);
}

// One wave iteration for all traversing lanes: majority vote between node steps and primitive steps.
// Returns (per lane) false when that lane's ray has just finished.  Lanes not voted for return true unchanged.
template <bool COUNT>
__device__ __forceinline__ bool travVote(const SceneView& sc, Trav& t, bool active, LaneStack& stack, TraceCounters& cnt) {
    const bool wantsPrim = active && travAtLeaf(t);
    const bool wantsNode = active && !travAtLeaf(t);
    const int nPrim = __popcll(__ballot(wantsPrim));
    const int nNode = __popcll(__ballot(wantsNode));
    bool more = true;
    if (nNode >= nPrim) {
        if (wantsNode) more = travNodeStep<COUNT>(sc, t, stack, cnt);
    } else {
        if (wantsPrim) more = travPrimStep<COUNT>(sc, t, stack, cnt);
    }
    return more;
}

// Whole-ray loop for one lane (ray-batch queries, MNEE chains): closest hit (ANY = false) or first hit
// (ANY = true).  Returns hit.prim == kHitMiss on a miss.  Same step functions as the persistent kernels.
template <bool ANY, bool COUNT>
__device__ __forceinline__ TraceHit traverse(const SceneView& sc, f3 org, f3 dir, float tnear, float tfar,
                                             LaneStack& stack, TraceCounters& cnt) {
    Trav t;
    if (!travBegin(sc, t, org, dir, tnear, tfar, ANY, stack)) return t.hit;
    bool more = true;
    while (more) {
        more = travAtLeaf(t) ? travPrimStep<COUNT>(sc, t, stack, cnt) : travNodeStep<COUNT>(sc, t, stack, cnt);
    }
    return t.hit;
}

}  // namespace ptrk
