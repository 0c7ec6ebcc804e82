// Two rays per lane: the step loop of the persistent traversal kernels (k_extend, k_connect).
//
// A wave executes one KIND of step per iteration for all 64 lanes - a node step (one two-box node) or a primitive
// step (one triangle / sphere) - chosen by vote, so the branch is wave-uniform.  With one ray per lane a lane idles
// whenever its ray is of the other kind or has finished (measured in round 1: node steps 55 % full, primitive steps
// 32 % full, VALU lane utilisation 0.50 - profiles/r1_pmc_counters.txt).  Here every lane holds TWO rays in registers,
// "A" and "B".  Only A is ever stepped; when A is not of the voted kind but B is, the lane exchanges the two register
// sets (18 v_swap_b32) and carries on with the other ray.  A lane idles only when NEITHER of its rays wants the voted
// kind, and a finished ray leaves its lane working on the other one until the next refill pass, which can therefore
// wait until many ray registers are empty (one batch of loads, no stall per finished ray).
//
// This is compaction in time instead of space: CDNA issues a wave64 instruction over all 64 lanes whatever the exec
// mask, so moving live rays to neighbouring lanes of the same wave (ds_bpermute) buys nothing; what pays is giving
// each lane a second ray to fall back on.  Re-binning rays ACROSS waves through LDS was costed at 2 x 18 LDS
// operations per ray per kind change (a ray changes kind ~4 times) plus a block-wide queue; the in-lane swap has the
// same effect on utilisation at 18 register moves and no synchronisation.
//
// Reference loop being replaced: shaders/pathtrace.metal:1971-2165 (one ray per thread, 128-entry private stack).
#pragma once

#include "traverse.h"

namespace ptrk {

constexpr uint32_t kRayIdle = 0xFFFFFFFFu;   // `cur` of a register set that holds no ray (never a valid node / leaf reference)

// One ray's registers.  Everything that has to follow the ray when the lane exchanges A and B is in here:
// 18 VGPRs.  tnear is the same constant for every ray of the persistent kernels (1e-4, the Embree path's epsilon).
struct RayRegs {
    f3 org, dir;
    f3 inv, oi;         // reciprocal direction / origin * reciprocal, in the space of the node boxes (traverse.h)
    float tfar;         // current closest distance (or the ray's length limit while nothing is hit)
    uint32_t prim;      // closest primitive so far (kHitMiss)
    uint32_t cur;       // internal node index, or the leaf reference of the primitives still to test; kRayIdle = no ray
    uint32_t sp;        // stack pointer (levels)
    uint32_t tag;       // caller's: path slot (k_extend) / record address (k_connect)
    uint32_t column;    // stack column of this ray: (which of the lane's two columns) | any-hit flag << 1
};

__device__ __forceinline__ bool rayIdle(const RayRegs& r) { return r.cur == kRayIdle; }
__device__ __forceinline__ bool rayAtLeaf(const RayRegs& r) { return (r.cur & kRefLeafBit) != 0u; }   // also true for kRayIdle: test idle first
__device__ __forceinline__ bool rayAnyHit(const RayRegs& r) { return (r.column & 2u) != 0u; }

#define PTR_SWAP32(x, y) asm volatile("v_swap_b32 %0, %1" : "+v"(x), "+v"(y))

__device__ __forceinline__ void swapRays(RayRegs& a, RayRegs& b) {
    PTR_SWAP32(a.org.x, b.org.x); PTR_SWAP32(a.org.y, b.org.y); PTR_SWAP32(a.org.z, b.org.z);
    PTR_SWAP32(a.dir.x, b.dir.x); PTR_SWAP32(a.dir.y, b.dir.y); PTR_SWAP32(a.dir.z, b.dir.z);
    PTR_SWAP32(a.inv.x, b.inv.x); PTR_SWAP32(a.inv.y, b.inv.y); PTR_SWAP32(a.inv.z, b.inv.z);
    PTR_SWAP32(a.oi.x, b.oi.x); PTR_SWAP32(a.oi.y, b.oi.y); PTR_SWAP32(a.oi.z, b.oi.z);
    PTR_SWAP32(a.tfar, b.tfar);
    PTR_SWAP32(a.prim, b.prim);
    PTR_SWAP32(a.cur, b.cur);
    PTR_SWAP32(a.sp, b.sp);
    PTR_SWAP32(a.tag, b.tag);
    PTR_SWAP32(a.column, b.column);
}

// Stack of the ray being stepped: lds[(level * 2 + column) * kTraceBlock + lane]; levels >= kDualLdsLevels live in a
// lane-interleaved HBM area (the builder bounds the tree depth, so it cannot overflow).
struct DualStack {
    LdsWord* lds;         // &ldsStack[threadIdx.x]
    uint32_t* spill;      // spill area of this launch
    uint32_t spillStride; // words per spill level: 2 * threads of the launch
    __device__ __forceinline__ uint32_t* spillSlot(uint32_t level, uint32_t column) const {
        const uint32_t thread = blockIdx.x * kTraceBlock + threadIdx.x;
        return spill + static_cast<size_t>(level - kDualLdsLevels) * spillStride + thread * 2u + (column & 1u);
    }
    __device__ __forceinline__ void push(RayRegs& r, uint32_t v) const {
        if (r.sp < kDualLdsLevels) {
            lds[(r.sp * 2u + (r.column & 1u)) * kTraceBlock] = v;
        } else if (r.sp < kTraversalStackDepth) {
            *spillSlot(r.sp, r.column) = v;
        } else {
            return;
        }
        ++r.sp;
    }
    __device__ __forceinline__ uint32_t pop(RayRegs& r) const {
        --r.sp;
        return (r.sp < kDualLdsLevels) ? lds[(r.sp * 2u + (r.column & 1u)) * kTraceBlock] : *spillSlot(r.sp, r.column);
    }
};

// Start a ray in register set `r` (its stack column is kept).  Returns false when the scene is empty.
template <bool QUANT>
__device__ __forceinline__ bool rayBegin(const SceneView& sc, RayRegs& r, f3 org, f3 dir, float tfar, bool anyHit, uint32_t tag, const DualStack& stack) {
    r.org = org;
    r.dir = dir;
    constexpr float kInvMax = 1.0e28f;   // see travBegin (traverse.h): a finite reciprocal keeps the fma slab test free of NaNs
    r.inv = mk3(fminf(fmaxf(__builtin_amdgcn_rcpf(dir.x), -kInvMax), kInvMax), fminf(fmaxf(__builtin_amdgcn_rcpf(dir.y), -kInvMax), kInvMax),
                fminf(fmaxf(__builtin_amdgcn_rcpf(dir.z), -kInvMax), kInvMax));
    if (QUANT) {
        const f3 cell = mk3(sc.gridCell[0], sc.gridCell[1], sc.gridCell[2]);
        const f3 invCell = mk3(sc.gridInvCell[0], sc.gridInvCell[1], sc.gridInvCell[2]);
        const f3 orgQ = (org - mk3(sc.gridOrigin[0], sc.gridOrigin[1], sc.gridOrigin[2])) * invCell;
        r.inv = r.inv * cell;
        r.oi = orgQ * r.inv;
    } else {
        r.oi = org * r.inv;
    }
    r.tfar = tfar;
    r.prim = kHitMiss;
    r.sp = 0u;
    r.tag = tag;
    r.column = (r.column & 1u) | (anyHit ? 2u : 0u);
    r.cur = sc.rootRef;   // kRefEmpty == kRayIdle: an empty scene leaves the set idle
    if (sc.oversizeRef != kRefEmpty) {
        // triangles kept out of the tree first (see travBegin)
        if (sc.rootRef != kRefEmpty) stack.push(r, sc.rootRef);
        r.cur = sc.oversizeRef;
        return true;
    }
    return sc.rootRef != kRefEmpty;
}

// pops the next subtree into r.cur; false (and r.cur = kRayIdle) when the traversal is complete
__device__ __forceinline__ bool rayPop(RayRegs& r, const DualStack& stack) {
    if (r.sp == 0u) {
        r.cur = kRayIdle;
        return false;
    }
    r.cur = stack.pop(r);
    return true;
}

// Node step of ray r (r.cur is an internal node): both child boxes of the node tested, nearer child next, the other
// one pushed.  Same arithmetic as travNodeStep (traverse.h); the node format is a compile-time choice here.
template <bool QUANT, bool COUNT>
__device__ __forceinline__ bool rayNodeStep(const SceneMem& mem, RayRegs& r, float tnear, const DualStack& stack, TraceCounters& cnt) {
    uint32_t ref0, ref1;
    float e0, e1;
    bool h0, h1;
    if (QUANT) {
        const uint4 q0 = load16u(mem.nodes, r.cur * 32u), q1 = load16u(mem.nodes, r.cur * 32u + 16u);
        ref0 = q0.w;
        ref1 = q1.w;
        h0 = slabTest(gridLo(q0.x, q0.y), gridHi(q0.y, q0.z), r.oi, r.inv, tnear, r.tfar, e0);
        h1 = slabTest(gridLo(q1.x, q1.y), gridHi(q1.y, q1.z), r.oi, r.inv, tnear, r.tfar, e1);
    } else {
        const uint32_t at = r.cur * 64u;
        const float4 n0 = load16f(mem.nodes, at), n1 = load16f(mem.nodes, at + 16u), n2 = load16f(mem.nodes, at + 32u),
                     n3 = load16f(mem.nodes, at + 48u);
        ref0 = __float_as_uint(n0.w);
        ref1 = __float_as_uint(n1.w);
        h0 = slabTest(mk3(n0), mk3(n1), r.oi, r.inv, tnear, r.tfar, e0);
        h1 = slabTest(mk3(n2), mk3(n3), r.oi, r.inv, tnear, r.tfar, e1);
    }
    h0 = h0 & (ref0 != kRefEmpty);
    h1 = h1 & (ref1 != kRefEmpty);
    if (COUNT) ++cnt.nodes;
    const bool firstIs0 = e0 <= e1;
    const uint32_t nearRef = (h0 & (firstIs0 | !h1)) ? ref0 : ref1;
    const uint32_t farRef = firstIs0 ? ref1 : ref0;
    if (h0 & h1) stack.push(r, farRef);
    if (h0 | h1) {
        r.cur = nearRef;
        return true;
    }
    return rayPop(r, stack);
}

// Primitive step of ray r (r.cur is a leaf reference): tests the leaf's first remaining primitive and advances the
// reference in place (first + 1, count - 1), so no separate position register travels with the ray.
template <bool COUNT>
__device__ __forceinline__ bool rayPrimStep(const SceneView& sc, const SceneMem& mem, RayRegs& r, float tnear, const DualStack& stack,
                                            TraceCounters& cnt) {
    const uint32_t cur = r.cur;
    const uint32_t index = cur & kRefOffsetMask;
    const uint32_t left = (cur >> kRefCountShift) & 0xFu;   // primitives after this one
    if (COUNT) {
        // counting build: bit 2 of `column` remembers that the ray is inside a leaf it has already been counted for
        ++cnt.prims;
        if (!(r.column & 4u)) { ++cnt.nodes; ++cnt.leaves; }
        r.column = left != 0u ? (r.column | 4u) : (r.column & ~4u);
    }
    bool found = false;
    if (cur & kRefSphereBit) {
        float tt;
        if (sphereTest(sc.spheres[index], r.org, r.dir, tnear, r.tfar, tt)) {
            r.tfar = tt;
            r.prim = index | kHitSphereBit;
            found = true;
        }
    } else {
        const uint32_t at = index * 48u;
        const float4 a = load16f(mem.tris, at), b = load16f(mem.tris, at + 16u), c = load16f(mem.tris, at + 32u);
        float tt, u, v;
        if (triangleTest(mk3(a), mk3(b), mk3(c), r.org, r.dir, tnear, r.tfar, tt, u, v)) {
            r.tfar = tt;
            r.prim = index;
            found = true;
        }
    }
    if (found && rayAnyHit(r)) {
        r.cur = kRayIdle;
        return false;
    }
    if (left != 0u) {
        r.cur = cur + 1u - (1u << kRefCountShift);
        return true;
    }
    return rayPop(r, stack);
}

}  // namespace ptrk
