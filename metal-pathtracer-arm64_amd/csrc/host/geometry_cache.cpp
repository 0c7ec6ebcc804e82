#include "geometry_cache.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace ptr {

namespace {

constexpr uint64_t kMagic = 0x3330454743525450ull;   // "PTRCGE03" (02: unused places of a wide node hold an inverted box; 03: wide nodes collapsed by area, their depth on record)

struct Header {
    uint64_t magic, fingerprint;
    uint64_t nodes, qnodes, triOrder, sphereOrder, triData, triNormals, sphereData, sphereInfo, triUv, triTangent, rectTriLeaf, wideWords;   // element counts
    float gridOrigin[3], gridCell[3], meanPrimExtent;
    uint32_t rootRef, oversizeRef, nodeCount, leafCount, maxDepth, maxLeafSize, triCount, sphereCount, useQuantized, wideCount, wideDepth, pad0;
    double sahCost, gatherSeconds, buildSeconds, flattenSeconds;
};

void fnv(uint64_t& h, const void* data, size_t bytes) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
}

// head, middle and tail of an array: enough to tell two meshes of the same size apart without hashing gigabytes
void fnvSampled(uint64_t& h, const void* data, size_t bytes) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    if (!p) return;
    const size_t piece = 4096;
    if (bytes <= 3 * piece) {
        fnv(h, p, bytes);
        return;
    }
    fnv(h, p, piece);
    fnv(h, p + (bytes / 2 / 16) * 16, piece);
    fnv(h, p + bytes - piece, piece);
}

template <typename T>
bool put(std::FILE* f, const T* data, uint64_t count) {
    return count == 0 || std::fwrite(data, sizeof(T), count, f) == count;
}

template <typename T>
bool get(std::FILE* f, std::vector<T>& v, uint64_t count) {
    v.resize(count);
    return count == 0 || std::fread(v.data(), sizeof(T), count, f) == count;
}

}  // namespace

uint64_t SceneFingerprint(const PtrSceneDesc& desc) {
    uint64_t h = 0xcbf29ce484222325ull;
    fnv(h, &desc.meshCount, sizeof(desc.meshCount));
    for (uint32_t m = 0; m < desc.meshCount; ++m) {
        const PtrMeshDesc& mesh = desc.meshes[m];
        fnv(h, &mesh.vertexCount, sizeof(mesh.vertexCount));
        fnv(h, &mesh.indexCount, sizeof(mesh.indexCount));
        fnv(h, mesh.localToWorld, sizeof(mesh.localToWorld));
        fnv(h, &mesh.materialIndex, sizeof(mesh.materialIndex));
        fnvSampled(h, mesh.positions, static_cast<size_t>(mesh.vertexCount) * 12u);
        fnvSampled(h, mesh.normals, static_cast<size_t>(mesh.vertexCount) * 12u);
        fnvSampled(h, mesh.indices, static_cast<size_t>(mesh.indexCount) * 4u);
        const uint32_t has = (mesh.uv0 ? 1u : 0u) | (mesh.uv1 ? 2u : 0u) | (mesh.tangents ? 4u : 0u);
        fnv(h, &has, sizeof(has));
        fnvSampled(h, mesh.uv0, static_cast<size_t>(mesh.vertexCount) * 8u);
        fnvSampled(h, mesh.uv1, static_cast<size_t>(mesh.vertexCount) * 8u);
        fnvSampled(h, mesh.tangents, static_cast<size_t>(mesh.vertexCount) * 16u);
    }
    fnv(h, &desc.rectCount, sizeof(desc.rectCount));
    if (desc.rects) fnv(h, desc.rects, static_cast<size_t>(desc.rectCount) * sizeof(PtrRect));
    fnv(h, &desc.sphereCount, sizeof(desc.sphereCount));
    if (desc.spheres) fnv(h, desc.spheres, static_cast<size_t>(desc.sphereCount) * sizeof(PtrSphere));
    fnv(h, &desc.materialCount, sizeof(desc.materialCount));
    for (uint32_t i = 0; i < desc.materialCount; ++i) fnv(h, desc.materials[i].typeEta, sizeof(float));   // the shade keys in the triangle records
    fnv(h, &desc.textureCount, sizeof(desc.textureCount));   // (decides whether the texture attributes are laid out)
    return h;
}

bool WriteGeometryCache(const std::string& path, const PreparedGeometry& pg, uint64_t fingerprint, std::string& error) {
    const SceneGeometry& g = pg.geo;
    const FlatBvh& b = g.bvh;
    Header hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = kMagic;
    hd.fingerprint = fingerprint;
    hd.nodes = b.nodes.size();
    hd.qnodes = b.qnodes.size();
    hd.triOrder = b.triOrder.size();
    hd.sphereOrder = b.sphereOrder.size();
    hd.triData = g.triData.size();
    hd.triNormals = g.triNormals.size();
    hd.sphereData = g.sphereData.size();
    hd.sphereInfo = g.sphereInfo.size();
    hd.triUv = g.triUv.size();
    hd.triTangent = g.triTangent.size();
    hd.rectTriLeaf = g.rectTriLeaf.size();
    hd.wideWords = static_cast<uint64_t>(pg.wideCount) * 16u;
    std::memcpy(hd.gridOrigin, b.gridOrigin, sizeof(hd.gridOrigin));
    std::memcpy(hd.gridCell, b.gridCell, sizeof(hd.gridCell));
    hd.meanPrimExtent = b.meanPrimExtent;
    hd.rootRef = b.rootRef;
    hd.oversizeRef = b.oversizeRef;
    hd.nodeCount = b.nodeCount;
    hd.leafCount = b.leafCount;
    hd.maxDepth = b.maxDepth;
    hd.maxLeafSize = b.maxLeafSize;
    hd.triCount = g.triCount;
    hd.sphereCount = g.sphereCount;
    hd.useQuantized = pg.useQuantized ? 1u : 0u;
    hd.wideCount = pg.wideCount;
    hd.wideDepth = pg.wideDepth;
    hd.pad0 = 0u;
    hd.sahCost = b.sahCost;
    hd.gatherSeconds = g.gatherSeconds;
    hd.buildSeconds = g.buildSeconds;
    hd.flattenSeconds = g.flattenSeconds;
    // written under a temporary name and renamed: a reader never sees half a file
    const std::string tmp = path + ".tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) {
        error = "cannot write geometry cache " + tmp;
        return false;
    }
    const bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1 && put(f, b.nodes.data(), hd.nodes) && put(f, b.qnodes.data(), hd.qnodes) &&
                    put(f, b.triOrder.data(), hd.triOrder) && put(f, b.sphereOrder.data(), hd.sphereOrder) && put(f, g.triData.data(), hd.triData) &&
                    put(f, g.triNormals.data(), hd.triNormals) && put(f, g.sphereData.data(), hd.sphereData) && put(f, g.sphereInfo.data(), hd.sphereInfo) &&
                    put(f, g.triUv.data(), hd.triUv) && put(f, g.triTangent.data(), hd.triTangent) && put(f, g.rectTriLeaf.data(), hd.rectTriLeaf) &&
                    put(f, pg.wide.get(), hd.wideWords);
    const bool closed = std::fclose(f) == 0;
    if (!ok || !closed || std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        error = "failed writing geometry cache " + path;
        return false;
    }
    return true;
}

bool ReadGeometryCache(const std::string& path, uint64_t fingerprint, PreparedGeometry& pg, std::string& error) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        error = "cannot open geometry cache " + path;
        return false;
    }
    Header hd;
    bool ok = std::fread(&hd, sizeof(hd), 1, f) == 1;
    if (ok && (hd.magic != kMagic || hd.fingerprint != fingerprint)) {
        std::fclose(f);
        error = "geometry cache " + path + " was built from another scene description";
        return false;
    }
    // element counts are bounded by what the builder can produce (64 M primitives): a corrupt header cannot ask for petabytes
    const uint64_t cap = 1ull << 31;
    ok = ok && hd.nodes < cap * 2 && hd.qnodes < cap && hd.triOrder < cap && hd.sphereOrder < cap && hd.triData < cap && hd.triNormals < cap &&
         hd.sphereData < cap && hd.sphereInfo < cap && hd.triUv < cap * 2 && hd.triTangent < cap && hd.rectTriLeaf < cap && hd.wideWords < cap;
    SceneGeometry& g = pg.geo;
    FlatBvh& b = g.bvh;
    ok = ok && get(f, b.nodes, hd.nodes) && get(f, b.qnodes, hd.qnodes) && get(f, b.triOrder, hd.triOrder) && get(f, b.sphereOrder, hd.sphereOrder) &&
         get(f, g.triData, hd.triData) && get(f, g.triNormals, hd.triNormals) && get(f, g.sphereData, hd.sphereData) && get(f, g.sphereInfo, hd.sphereInfo) &&
         get(f, g.triUv, hd.triUv) && get(f, g.triTangent, hd.triTangent) && get(f, g.rectTriLeaf, hd.rectTriLeaf);
    if (ok && hd.wideWords > 0) {
        pg.wide.reset(new uint32_t[hd.wideWords]);
        ok = std::fread(pg.wide.get(), sizeof(uint32_t), hd.wideWords, f) == hd.wideWords;
    }
    std::fclose(f);
    if (!ok) {
        error = "geometry cache " + path + " is truncated or corrupt";
        return false;
    }
    std::memcpy(b.gridOrigin, hd.gridOrigin, sizeof(hd.gridOrigin));
    std::memcpy(b.gridCell, hd.gridCell, sizeof(hd.gridCell));
    b.meanPrimExtent = hd.meanPrimExtent;
    b.rootRef = hd.rootRef;
    b.oversizeRef = hd.oversizeRef;
    b.nodeCount = hd.nodeCount;
    b.leafCount = hd.leafCount;
    b.maxDepth = hd.maxDepth;
    b.maxLeafSize = hd.maxLeafSize;
    b.sahCost = hd.sahCost;
    g.triCount = hd.triCount;
    g.sphereCount = hd.sphereCount;
    g.gatherSeconds = hd.gatherSeconds;
    g.buildSeconds = hd.buildSeconds;
    g.flattenSeconds = hd.flattenSeconds;
    pg.useQuantized = hd.useQuantized != 0u;
    pg.wideCount = hd.wideCount;
    pg.wideDepth = hd.wideDepth;
    return true;
}

}  // namespace ptr
