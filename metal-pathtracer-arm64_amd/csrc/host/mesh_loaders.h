// Mesh file readers feeding SceneResources::addMesh.
// Behaviour follows the reference's LoadObjMesh / LoadPlyMesh (src/renderer/SceneManager.mm:96-209, 223-518);
// the parsers themselves are written here (the reference delegates to tinyobjloader / tinyply).
#pragma once

#include <string>
#include <vector>

#include "scene_resources.h"

namespace ptr {

struct LoadedMeshData {
    std::vector<SceneResources::MeshVertex> vertices;
    std::vector<uint32_t> indices;
};

bool LoadObjMesh(const std::string& path, LoadedMeshData& out, std::string& error);
bool LoadPlyMesh(const std::string& path, LoadedMeshData& out, std::string& error);

}  // namespace ptr
