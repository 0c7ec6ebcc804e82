// Host-side SAH BVH construction and flattening to the device node layout (kernels/device_types.h).
//
// Takes the place of the reference's SoftwareBvhAccel::rebuild (src/renderer/SceneAccel.mm:23-325: per-mesh
// tinybvh binned-SAH BLAS + median-split TLAS) and BvhBuilder (src/renderer/BvhBuilder.mm, spheres).  The
// oracle path (Embree) bakes every mesh to world space and finds the true closest hit over one scene, so
// this builder does the same: ONE binned-SAH tree over all world-space triangles, rectangle halves and
// spheres.  Differences from the reference layout, all MI355X-motivated:
//   * 64 B nodes hold both child boxes + child references (one fetch decides both children),
//   * leaves are folded into the parent's child reference (no leaf node fetch),
//   * primitives are re-ordered into leaf order so a leaf is one contiguous run,
//   * tree depth is bounded (< kMaxTreeDepth) so the traversal stack can never overflow.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

namespace ptr {

struct BuildPrim {
    float lo[3], hi[3];   // bounds
    uint32_t isSphere;
};

struct FlatBvh {
    std::vector<float> nodes;          // 16 floats per node (see device_types.h)
    std::vector<uint32_t> qnodes;      // 8 words per node: 16-bit grid boxes + child refs (see device_types.h)
    float gridOrigin[3] = {0, 0, 0};
    float gridCell[3] = {1, 1, 1};
    float meanPrimExtent = 0.0f;       // mean of the in-tree primitives' largest box edge (quantisation quality gate)
    std::vector<uint32_t> triOrder;    // leaf-order -> index into the triangle input
    std::vector<uint32_t> sphereOrder; // leaf-order -> index into the sphere input
    uint32_t rootRef = 0xFFFFFFFFu;
    // Leaf reference of the triangles kept OUT of the tree (kRefEmpty: none): the few primitives so much larger than the rest
    // that, inside the tree, they would stretch the 16-bit grid of the quantised nodes over mostly empty space (the floor of
    // a room around a 28 M-triangle statue).  They are the last entries of triOrder and every ray tests them first.
    uint32_t oversizeRef = 0xFFFFFFFFu;
    uint32_t nodeCount = 0, leafCount = 0, maxDepth = 0, maxLeafSize = 0;
    double sahCost = 0.0;
};

// prims: triangles and spheres mixed (isSphere flag); indices in triOrder/sphereOrder refer to the n-th
// triangle / n-th sphere of the input in input order.
void BuildFlatBvh(const std::vector<BuildPrim>& prims, FlatBvh& out, uint32_t threads = 0, uint32_t leafMax = 4);

// Four-wide nodes for the persistent traversal kernels (kernels/traverse.h travWideStep).  A wide node is 16 words: four 16 B child
// records (quantised box + reference, as in FlatBvh::qnodes); unused places hold an inverted box and kRefEmpty.  The wide node rooted
// at a binary node keeps as children
//   ByArea:   what is left after opening, again and again, the internal child with the largest box until four children stand - the
//             children a ray is most likely to enter anyway are the ones opened, small ones stay closed and are culled whole (counted on
//             the BASELINE scenes' own trees, profiles/r3_wide_walk_counts.txt: 10-17 % fewer steps per ray than ByLevel);
//   ByLevel:  the children's children (every second level of the binary tree collapsed; a child that is a leaf keeps its record).
// Wide nodes are numbered in the order of the binary (preorder) indices of their roots, internal references renumbered to match.
// Returns the number of wide nodes; depthOut (nullable): levels of the wide tree (a step pushes at most three entries per level).
enum class WideCollapse { ByArea, ByLevel };
uint32_t BuildWideNodes(const FlatBvh& bvh, WideCollapse how, std::unique_ptr<uint32_t[]>& wide, uint32_t* depthOut = nullptr);

}  // namespace ptr
