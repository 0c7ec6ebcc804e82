// glTF 2.0 / GLB loader: geometry, metallic-roughness materials and their textures.  The Embree-parity integrator reads the
// factors only (textures are never sampled on that path); the Metal-semantics PBR model
// (PTR_METAL_PBR) samples the textures the loader decodes (PNG / baseline JPEG, csrc/host/image_decoders.h).  Interface follows the reference's
// include/assets/GltfLoader.h; behaviour follows src/assets/GltfLoader.mm:650-788, 792-1535.
#pragma once

#include <string>
#include <vector>

#include "scene_resources.h"

namespace ptr {

struct GltfCameraInfo {
    bool valid = false;
    bool hasPerspective = false;
    float yfov = 0.0f;
    float3 position{0.0f, 0.0f, 0.0f};
    float3 forward{0.0f, 0.0f, -1.0f};
    bool hasSceneBounds = false;
    float sceneRadius = 0.0f;
};

struct GltfLoadOptions {
    float emissiveScale = 1.0f;
    bool loadTextures = true;           // decode and register material textures
    bool forceLinearBaseColor = false;  // GltfLoadOptions of the reference: treat the base colour / emissive images as linear data
    bool forceLinearEmissive = false;
};

bool LoadGltfScene(const std::string& path, SceneResources& resources, std::string& error,
                   GltfCameraInfo* outCamera = nullptr, const GltfLoadOptions* options = nullptr);

}  // namespace ptr
