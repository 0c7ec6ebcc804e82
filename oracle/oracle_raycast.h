// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md.
//
// CPU ray caster standing in for the reference's Embree 4.4.0 calls (rtcIntersect1 / rtcOccluded1,
// src/headless/EmbreeHeadlessRenderer.mm:2302-2325, 2412-2433).  Embree itself is vendored in the reference
// (external/embree) but needs CMake + generated config headers, so it is treated as unbuildable here; this
// file restates the published semantics of the two primitive intersectors the reference's scene uses:
//   * triangles: external/embree/kernels/geometry/triangle_intersector_moeller.h:72-115
//     (e1 = v0-v1, e2 = v2-v0, Ng = e2 x e1; den != 0, U >= 0, V >= 0, U+V <= |den|;
//      |den|*tnear < T <= |den|*tfar; t,u,v = T,U,V / |den|)
//   * spheres (RTC_GEOMETRY_TYPE_SPHERE_POINT): external/embree/kernels/geometry/sphere_intersector.h:80-121
//     (front root if tnear <= t <= tfar, else back root)
// Scene assembly (world-space baked meshes, rectangles as two triangles with winding fixed to the stored
// normal, one closest hit over everything) follows BuildEmbreeScene, EmbreeHeadlessRenderer.mm:2077-2300.
// The acceleration structure is a plain binned-SAH BVH; results equal a brute-force loop over all
// primitives (checked in tests/test_oracle.py), so it does not influence what the oracle returns.
#pragma once

#include <cstdint>
#include <vector>

#include "oracle_math.h"
#include "ptr_abi.h"

namespace oracle {

enum class GeomType : uint32_t { Mesh = 0, Spheres = 1, Rectangles = 2 };

struct Prim {            // one entry per intersectable primitive
    V3 v0, e1, e2;       // triangle: v0, v0-v1, v2-v0        sphere: v0 = centre, e1.x = radius
    uint32_t geom;       // index into Scene::geoms
    uint32_t primId;     // triangle index inside its geometry / sphere index
    uint32_t isSphere;
};

struct Geom {
    GeomType type = GeomType::Mesh;
    uint32_t meshIndex = 0;             // for Mesh
    uint32_t materialIndex = 0;         // for Mesh
    std::vector<V3> normals;            // world-space vertex normals (Mesh) / per-vertex rect normals
    std::vector<uint32_t> indices;      // triangle indices
    std::vector<uint32_t> primMaterial; // Rectangles: per triangle; Spheres: per sphere
    std::vector<uint32_t> triToRect;    // Rectangles: triangle -> rectangle index
    // textured metallic-roughness model (Metal semantics only): per-vertex texture coordinates / world-space tangents, world positions
    std::vector<V3> positions;          // world space (Mesh)
    std::vector<float> uv0, uv1;        // 2 per vertex (empty: none)
    std::vector<float> tangents;        // 4 per vertex: world-space tangent (unnormalised), w = handedness x sign of det(localToWorld); empty: none
    float detSign = 1.0f;
};

struct RayHit {
    float t = -1.0f;     // < 0: miss
    float u = 0.0f, v = 0.0f;
    V3 ng;               // unnormalised geometric normal
    uint32_t geom = 0;
    uint32_t primId = 0;
};

struct Counters {
    uint64_t nodes = 0, prims = 0;
};

struct BvhNode {
    float lo[3], hi[3];
    uint32_t left;       // internal: left child index (right = left+1); leaf: first prim
    uint32_t count;      // 0 = internal
};

class Scene {
public:
    void build(const PtrSceneDesc& desc);

    // closest hit in (tnear, tfar] (triangles) / [tnear, tfar] (spheres)
    bool intersect(V3 org, V3 dir, float tnear, float tfar, RayHit& hit, bool bruteForce = false,
                   Counters* counters = nullptr) const;
    bool occluded(V3 org, V3 dir, float tnear, float tfar, bool bruteForce = false,
                  Counters* counters = nullptr) const;

    std::vector<Geom> geoms;
    std::vector<Prim> prims;            // in BVH leaf order
    std::vector<BvhNode> nodes;
    std::vector<PtrSphere> spheres;
    const PtrRect* rects = nullptr;
    uint32_t rectCount = 0;

private:
    void buildBvh();
};

}  // namespace oracle
