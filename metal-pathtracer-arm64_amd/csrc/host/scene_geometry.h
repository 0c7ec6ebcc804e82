// Host-side geometry preparation for the device scene: bake meshes to world space, split rectangles into two
// triangles, build the BVH and lay triangles / spheres out in leaf order.  Plain C++ (no HIP), so the CPU test
// suite can build and validate the exact arrays the GPU traverses.
//
// Reference counterparts: the Embree backend's scene assembly (src/headless/EmbreeHeadlessRenderer.mm:2100-2166
// meshes, 2184-2196 spheres, 2211-2293 rectangles) and SceneResources::rebuildAccelerationStructures
// (src/renderer/SceneResources.mm:2055-2259).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bvh_builder.h"
#include "ptr_abi.h"

namespace ptr {

struct SceneGeometry {
    FlatBvh bvh;
    std::vector<float> triData;      // 12 floats per triangle in leaf order: v0|material, v0-v1|kind<<30|geom, v2-v0|primIndex
    std::vector<float> triNormals;   // 12 floats per triangle in leaf order: world-space vertex normals
    std::vector<float> sphereData;   // 4 floats per sphere in leaf order: centre, radius
    std::vector<uint32_t> sphereInfo;  // 2 words per sphere in leaf order: original index, material
    // textured scenes only (desc.textureCount > 0 and some mesh carries texture coordinates); leaf order:
    std::vector<float> triUv;        // 16 floats per triangle: (uv0, uv1) of the three vertices, then (uvPerWorld0, uvPerWorld1, sign of det(localToWorld), 0)
    std::vector<float> triTangent;   // 12 floats per triangle: world-space vertex tangents, w = handedness x sign of det (0: no tangent)
    std::vector<uint32_t> rectTriLeaf;   // 2 per rectangle: leaf-order indices of its two triangles
    uint32_t triCount = 0, sphereCount = 0;
    double gatherSeconds = 0.0, buildSeconds = 0.0, flattenSeconds = 0.0;
};

// leafMax = 0 picks the default (4).  Returns false with a message on malformed input.
bool BuildSceneGeometry(const PtrSceneDesc& desc, uint32_t leafMax, SceneGeometry& out, std::string& error);

struct GeometryCheck {
    uint64_t nodes = 0, leaves = 0, trianglesReferenced = 0, spheresReferenced = 0;
    uint64_t maxDepth = 0, maxLeafSize = 0;
    uint64_t unreferenced = 0;        // primitives no leaf points at
    uint64_t multiplyReferenced = 0;  // primitives in more than one leaf
    uint64_t boxViolations = 0;       // primitives sticking out of an ancestor's child box (float nodes)
    uint64_t quantViolations = 0;     // float child boxes sticking out of their quantised twin
    uint64_t badRefs = 0;             // child references pointing outside the arrays / cycles
    uint64_t oversize = 0;            // triangles kept out of the tree (FlatBvh::oversizeRef)
    uint64_t wideNodes = 0;           // four-wide nodes (BuildWideNodes, by area)
    uint64_t wideDepth = 0;           // levels of that wide tree
    uint64_t wideProblems = 0;        // bad references in them + primitives not reached exactly once through them
};

// Walks the flattened tree from the root and checks the invariants the device traversal relies on.
void ValidateSceneGeometry(const SceneGeometry& g, GeometryCheck& out);

}  // namespace ptr
