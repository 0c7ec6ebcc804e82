#include "gltf_loader.h"

namespace ptr {

bool LoadGltfScene(const std::string& path, SceneResources&, std::string& error, GltfCameraInfo*,
                   const GltfLoadOptions*) {
    error = "glTF loading is not built yet: " + path;
    return false;
}

}  // namespace ptr
