// BVH reference encoding shared by the host builder (plain C++) and the device traversal.
#pragma once

#include <cstdint>

namespace ptrk {

// ---- BVH child reference encoding (one per child slot of a 64 B node) ----
//   0xFFFFFFFF                 : empty slot
//   bit31 = 1                  : leaf; bit30 = sphere leaf; bits 26..29 = primCount-1; bits 0..25 = first prim
//   otherwise                  : index of an internal node
constexpr uint32_t kRefEmpty = 0xFFFFFFFFu;
constexpr uint32_t kRefLeafBit = 0x80000000u;
constexpr uint32_t kRefSphereBit = 0x40000000u;
constexpr uint32_t kRefCountShift = 26u;
constexpr uint32_t kRefOffsetMask = 0x03FFFFFFu;
constexpr uint32_t kMaxLeafPrims = 8u;
constexpr uint32_t kMaxTreeDepth = 48u;          // the builder bounds the depth of the binary tree below this
// Entries a traversal stack can hold.  The binary walk pushes at most one entry per level (< kMaxTreeDepth); the four-wide walk
// (two binary levels per step) up to three per step: 3 x kMaxTreeDepth / 2 = 72, plus the root beside the oversize leaf.
constexpr uint32_t kTraversalStackDepth = 76u;
constexpr uint32_t kLdsStackLevels = 16u;        // stack levels kept in LDS; deeper levels spill to HBM
#ifndef PTR_TRACE_BLOCK   // 64, 128 or 256
#define PTR_TRACE_BLOCK 256
#endif
constexpr uint32_t kTraceBlock = PTR_TRACE_BLOCK;   // threads per block of the traversal kernels
constexpr uint32_t kTraceGridUnit = 256u;           // the host sizes traversal grids (and the spill area) in units of this many threads
// Hit word of a traced ray: kHitMiss, or  bit31 = sphere | bits 0..25 = leaf-order primitive index.
// Meta word of a triangle record (t[1].w): bits 31:30 = kind (0 mesh triangle, 2 rectangle half), bits 26..29 = material class
// (material type + 1, never 0), bits 0..25 = geometry index.  The class field makes a rectangle's meta word distinct from every hit word
// (a sphere hit word has bit 31 set and zeros in bits 26..30): an any-hit query that must ignore a rectangle's own triangles starts
// with that meta word in its hit word (wavefront.hip, kind-3 records; anyHitFound).
constexpr uint32_t kHitMiss = 0xFFFFFFFFu;
constexpr uint32_t kHitSphereBit = 0x80000000u;
constexpr uint32_t kHitKeyShift = 26u;
constexpr uint32_t kHitKeyMask = 0xFu;
constexpr uint32_t kHitIndexMask = kRefOffsetMask;
constexpr uint32_t kTriGeomMask = (1u << kHitKeyShift) - 1u;   // geometry index field of t[1].w

}  // namespace ptrk
