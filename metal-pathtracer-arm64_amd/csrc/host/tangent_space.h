// Tangent generation for meshes that carry texture coordinates but no tangents (glTF primitives without TANGENT).
// Reference: src/assets/TangentGen.mm:181-230 (de-index, MikkTSpace, angle-weighted fallback) over external/MikkTSpace/mikktspace.c.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "scene_resources.h"

namespace ptr {

// The MikkTSpace method on a triangle soup: corner c of triangle f is element 3 f + c of positions (xyz), normals (xyz, unit length)
// and uvs (st); tangents receives xyz + sign (+1 where the mapping preserves orientation, -1 where it mirrors) per corner.
// Returns false when there is nothing to do.  angularThresholdDegrees: 180 = genTangSpaceDefault.
bool GenerateTangentSpace(const float* positions, const float* normals, const float* uvs, size_t triangleCount, float* tangents,
                          float angularThresholdDegrees = 180.0f);

// What the reference does to a primitive without tangents: every triangle corner becomes a vertex of its own (indices 0, 1, 2, ...),
// and each gets the MikkTSpace tangent of its corner; when that fails, angle-weighted per-vertex tangents from the UV derivatives.
void GenerateTangents(std::vector<SceneResources::MeshVertex>& vertices, std::vector<uint32_t>& indices);

}  // namespace ptr
