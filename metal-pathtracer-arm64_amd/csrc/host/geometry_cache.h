// The device-independent geometry of a prepared scene - BVH, leaf-order triangle / sphere arrays, four-wide nodes - as one file, so
// that the processes of a multi-GPU render (one per device, bench.py --gpus N) build the BVH ONCE: the first process prepares the
// scene and writes the file (to /dev/shm: it never touches a disk), the others map it instead of running N builds side by side on
// one host (6.5 s each alone for the 29 M-triangle scene).  The file carries a fingerprint of the scene description it was built
// from; a reader with another description is refused.
#pragma once

#include <cstdint>
#include <memory>
#include <string>

#include "scene_geometry.h"

namespace ptr {

struct PreparedGeometry {
    SceneGeometry geo;
    bool useQuantized = false;           // node format of the persistent kernels (decided from the grid's cell size)
    std::unique_ptr<uint32_t[]> wide;    // four-wide nodes (16 words each) when the scene uses them
    uint32_t wideCount = 0;
    uint32_t wideDepth = 0;              // levels of the wide tree (BuildWideNodes): sizes the traversal stack
};

// what the geometry depends on: primitive counts, transforms, material types, and samples of the vertex / index data
uint64_t SceneFingerprint(const PtrSceneDesc& desc);

bool WriteGeometryCache(const std::string& path, const PreparedGeometry& pg, uint64_t fingerprint, std::string& error);
bool ReadGeometryCache(const std::string& path, uint64_t fingerprint, PreparedGeometry& pg, std::string& error);

}  // namespace ptr
