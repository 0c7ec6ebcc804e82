#include "scene_resources.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "hdr_loader.h"

namespace ptr {
namespace {

inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

inline float3 clampColor01(const float3& v) {
    return {clampf(v.x, 0.0f, 1.0f), clampf(v.y, 0.0f, 1.0f), clampf(v.z, 0.0f, 1.0f)};
}

inline float3 clampPositive(const float3& v) {
    return {std::max(v.x, 0.0f), std::max(v.y, 0.0f), std::max(v.z, 0.0f)};
}

inline void set4(float dst[4], const float3& v, float w) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
    dst[3] = w;
}

inline void set4(float dst[4], float x, float y, float z, float w) {
    dst[0] = x;
    dst[1] = y;
    dst[2] = z;
    dst[3] = w;
}

// Hemispherical average of Schlick Fresnel: F0 + (1-F0)/21 (SceneResources.mm:823-832).
float coatFresnelAverage(float coatIor) {
    const float eta = std::max(coatIor, 1.0f);
    const float ratio = (eta - 1.0f) / std::max(eta + 1.0f, 1.0e-6f);
    const float f0 = ratio * ratio;
    const float average = f0 + (1.0f - f0) * (1.0f / 21.0f);
    return clampf(average, 0.0f, 0.999f);
}

// SceneResources.mm:834-852
float coatSampleWeight(MaterialType type, float coatRoughness, float coatThickness, float coatAverage) {
    const bool hasLayer = (coatThickness > 1.0e-4f) || (coatRoughness > 1.0e-4f) ||
                          type == MaterialType::Plastic || type == MaterialType::CarPaint;
    if (!hasLayer) {
        return 0.0f;
    }
    float weight = coatAverage * 2.5f + coatRoughness * 0.5f;
    if (type == MaterialType::CarPaint) {
        weight = std::max(weight, 0.35f);
    } else if (type == MaterialType::Plastic) {
        weight = std::max(weight, 0.25f);
    }
    return clampf(weight, 0.0f, 0.95f);
}

void setTextureDefaults(PtrMaterial& m) {
    for (int i = 0; i < 4; ++i) {
        m.textureIndices0[i] = kInvalidTextureIndex;
        m.textureIndices1[i] = kInvalidTextureIndex;
        m.textureUvSet0[i] = 0u;
        m.textureUvSet1[i] = 0u;
    }
    for (int t = 0; t < 12; ++t) {
        // even rows (1,0,0,0), odd rows (0,1,0,0): identity 2x3 UV transforms
        set4(m.textureTransform[t], (t % 2 == 0) ? 1.0f : 0.0f, (t % 2 == 0) ? 0.0f : 1.0f, 0.0f, 0.0f);
    }
}

}  // namespace

void SceneResources::clear() {
    m_spheres.clear();
    m_rectangles.clear();
    m_materials.clear();
    m_materialNames.clear();
    m_meshes.clear();
    m_textures.clear();
    m_envRgba.clear();
    m_envWidth = 0;
    m_envHeight = 0;
}

uint32_t SceneResources::addMaterial(const MaterialParams& p) {
    if (m_materials.size() >= kMaxMaterials) {
        return kMaxMaterials - 1;
    }
    const uint32_t index = static_cast<uint32_t>(m_materials.size());

    PtrMaterial m;
    std::memset(&m, 0, sizeof(m));
    setTextureDefaults(m);

    const float roughness = clampf(p.roughness, 0.0f, 1.0f);
    set4(m.baseColorRoughness, clampColor01(p.baseColor), roughness);

    const float ior = std::max(p.indexOfRefraction, 0.0f);
    const float coatIor = std::max(p.coatIor, 0.0f);
    set4(m.typeEta, static_cast<float>(static_cast<uint32_t>(p.type)), ior, coatIor, p.thinDielectric ? 1.0f : 0.0f);
    set4(m.emission, p.emission, p.emissionUsesEnvironment ? 1.0f : 0.0f);

    const float conductorFlag = p.hasConductorParameters ? 1.0f : 0.0f;
    set4(m.conductorEta, clampPositive(p.conductorEta), conductorFlag);
    set4(m.conductorK, clampPositive(p.conductorK), conductorFlag);

    const float coatRoughness = clampf(p.coatRoughness, 0.0f, 1.0f);
    const float coatThickness = std::max(p.coatThickness, 0.0f);
    const float coatAverage = coatFresnelAverage(coatIor);
    const float coatWeight = clampf(coatSampleWeight(p.type, coatRoughness, coatThickness, coatAverage), 0.0f, 0.95f);
    set4(m.coatParams, coatRoughness, coatThickness, coatWeight, coatAverage);
    set4(m.coatTint, clampColor01(p.coatTint), 0.0f);
    set4(m.coatAbsorption, clampPositive(p.coatAbsorption), 0.0f);
    set4(m.dielectricSigmaA, clampPositive(p.dielectricSigmaA), 0.0f);

    set4(m.sssSigmaA, clampPositive(p.sssSigmaA), p.sssSigmaOverride ? 1.0f : 0.0f);
    set4(m.sssSigmaS, clampPositive(p.sssSigmaS), clampf(p.sssAnisotropy, -0.99f, 0.99f));
    set4(m.sssParams, std::max(p.sssMeanFreePath, 0.0f), static_cast<float>(p.sssMethod),
         p.sssCoatEnabled ? 1.0f : 0.0f, 0.0f);

    const float flakeReflectance = clampf(p.carpaintFlakeReflectanceScale, 0.0f, 1.0f);
    float flakeWeight = clampf(p.carpaintFlakeSampleWeight, 0.0f, 0.95f);
    // flake sampling weight follows the lobe's energy (SceneResources.mm:1001-1003)
    flakeWeight = clampf(flakeWeight * std::max(flakeReflectance, 0.01f), 0.0f, 0.95f);
    set4(m.carpaintBaseParams, clampf(p.carpaintBaseMetallic, 0.0f, 1.0f), clampf(p.carpaintBaseRoughness, 0.0f, 1.0f),
         std::max(p.carpaintFlakeScale, 1.0e-4f), flakeReflectance);
    set4(m.carpaintFlakeParams, flakeWeight, clampf(p.carpaintFlakeRoughness, 0.0f, 1.0f),
         clampf(p.carpaintFlakeAnisotropy, -0.99f, 0.99f), clampf(p.carpaintFlakeNormalStrength, 0.0f, 1.0f));

    float3 baseEta = clampPositive(p.carpaintBaseEta);
    float3 baseK = clampPositive(p.carpaintBaseK);
    if (!p.carpaintHasBaseConductor) {
        baseEta = {0.0f, 0.0f, 0.0f};
        baseK = {0.0f, 0.0f, 0.0f};
    }
    const float baseFlag = p.carpaintHasBaseConductor ? 1.0f : 0.0f;
    set4(m.carpaintBaseEta, baseEta, baseFlag);
    set4(m.carpaintBaseK, baseK, baseFlag);
    set4(m.carpaintBaseTint, clampColor01(p.carpaintBaseTint), 0.0f);
    set4(m.pbrParams, 0.0f, roughness, 1.0f, 1.0f);

    m_materials.push_back(m);
    m_materialNames.push_back(p.name.empty() ? ("Material " + std::to_string(index)) : p.name);
    return index;
}

uint32_t SceneResources::addMaterialData(const PtrMaterial& material, std::string name) {
    if (m_materials.size() >= kMaxMaterials) {
        return kMaxMaterials - 1;
    }
    const uint32_t index = static_cast<uint32_t>(m_materials.size());
    PtrMaterial m = material;
    for (int i = 0; i < 4; ++i) {
        m.textureUvSet0[i] = std::min<uint32_t>(m.textureUvSet0[i], 1u);
    }
    m.textureUvSet1[0] = std::min<uint32_t>(m.textureUvSet1[0], 1u);
    m.textureUvSet1[1] = std::min<uint32_t>(m.textureUvSet1[1], 1u);
    m_materials.push_back(m);
    m_materialNames.push_back(name.empty() ? ("Material " + std::to_string(index)) : std::move(name));
    return index;
}

void SceneResources::addSphere(const float3& center, float radius, uint32_t materialIndex) {
    if (m_spheres.size() >= kMaxSpheres) {
        return;
    }
    if (materialIndex >= materialCount()) {
        materialIndex = materialCount() == 0 ? 0u : materialCount() - 1;
    }
    PtrSphere s;
    std::memset(&s, 0, sizeof(s));
    set4(s.centerRadius, center, radius);
    s.materialIndex[0] = materialIndex;
    m_spheres.push_back(s);
}

void SceneResources::addRectangle(const float3& boundsMin, const float3& boundsMax, uint32_t normalAxis,
                                  bool normalPositive, bool twoSided, uint32_t materialIndex) {
    if (m_rectangles.size() >= kMaxRectangles) {
        return;
    }
    if (materialIndex >= materialCount()) {
        materialIndex = materialCount() == 0 ? 0u : materialCount() - 1;
    }
    normalAxis = std::min(normalAxis, 2u);

    const float3 lo{std::min(boundsMin.x, boundsMax.x), std::min(boundsMin.y, boundsMax.y),
                    std::min(boundsMin.z, boundsMax.z)};
    const float3 hi{std::max(boundsMin.x, boundsMax.x), std::max(boundsMin.y, boundsMax.y),
                    std::max(boundsMin.z, boundsMax.z)};

    // Corner/edge conventions of SceneResources.mm:1770-1817: edgeU x edgeV points along the requested
    // normal, and the negative-facing variants start from the opposite in-plane corner.
    float3 corner, edgeU, edgeV, desired;
    if (normalAxis == 0) {
        edgeU = {0.0f, hi.y - lo.y, 0.0f};
        if (normalPositive) {
            corner = {hi.x, lo.y, lo.z};
            edgeV = {0.0f, 0.0f, hi.z - lo.z};
        } else {
            corner = {lo.x, lo.y, hi.z};
            edgeV = {0.0f, 0.0f, lo.z - hi.z};
        }
        desired = {normalPositive ? 1.0f : -1.0f, 0.0f, 0.0f};
    } else if (normalAxis == 1) {
        edgeU = {hi.x - lo.x, 0.0f, 0.0f};
        if (normalPositive) {
            corner = {lo.x, hi.y, lo.z};
            edgeV = {0.0f, 0.0f, hi.z - lo.z};
        } else {
            corner = {lo.x, lo.y, hi.z};
            edgeV = {0.0f, 0.0f, lo.z - hi.z};
        }
        desired = {0.0f, normalPositive ? 1.0f : -1.0f, 0.0f};
    } else {
        edgeV = {0.0f, hi.y - lo.y, 0.0f};
        if (normalPositive) {
            corner = {lo.x, lo.y, hi.z};
            edgeU = {hi.x - lo.x, 0.0f, 0.0f};
        } else {
            corner = {hi.x, lo.y, lo.z};
            edgeU = {lo.x - hi.x, 0.0f, 0.0f};
        }
        desired = {0.0f, 0.0f, normalPositive ? 1.0f : -1.0f};
    }
    storeRectangleOriented(corner, edgeU, edgeV, twoSided, materialIndex, desired);
}

void SceneResources::addBox(const float3& minCorner, const float3& maxCorner, uint32_t materialIndex,
                            bool includeBottomFace, bool twoSided) {
    addBoxTransformed(minCorner, maxCorner, materialIndex, float4x4::identity(), includeBottomFace, twoSided);
}

void SceneResources::addBoxTransformed(const float3& minCorner, const float3& maxCorner, uint32_t materialIndex,
                                       const float4x4& transform, bool includeBottomFace, bool twoSided) {
    if (materialIndex >= materialCount()) {
        materialIndex = materialCount() == 0 ? 0u : materialCount() - 1;
    }
    const float3 lo{std::min(minCorner.x, maxCorner.x), std::min(minCorner.y, maxCorner.y),
                    std::min(minCorner.z, maxCorner.z)};
    const float3 hi{std::max(minCorner.x, maxCorner.x), std::max(minCorner.y, maxCorner.y),
                    std::max(minCorner.z, maxCorner.z)};
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;

    struct Face {
        float3 corner, edgeU, edgeV, normal;
        bool include;
    };
    // Face order +X,-X,+Y,-Y,+Z,-Z (SceneResources.mm:1867-1874); it fixes rectangle indices.
    const Face faces[6] = {
        {{hi.x, lo.y, lo.z}, {0, dy, 0}, {0, 0, dz}, {1, 0, 0}, true},
        {{lo.x, lo.y, hi.z}, {0, dy, 0}, {0, 0, -dz}, {-1, 0, 0}, true},
        {{lo.x, hi.y, lo.z}, {dx, 0, 0}, {0, 0, dz}, {0, 1, 0}, true},
        {{lo.x, lo.y, hi.z}, {dx, 0, 0}, {0, 0, -dz}, {0, -1, 0}, includeBottomFace},
        {{lo.x, lo.y, hi.z}, {dx, 0, 0}, {0, dy, 0}, {0, 0, 1}, true},
        {{hi.x, lo.y, lo.z}, {-dx, 0, 0}, {0, dy, 0}, {0, 0, -1}, true},
    };
    auto xformPoint = [&](const float3& p) {
        const float4 r = mul(transform, float4(p, 1.0f));
        return float3{r.x, r.y, r.z};
    };
    auto xformVector = [&](const float3& v) {
        const float4 r = mul(transform, float4(v, 0.0f));
        return float3{r.x, r.y, r.z};
    };
    for (const Face& f : faces) {
        if (!f.include) {
            continue;
        }
        storeRectangleOriented(xformPoint(f.corner), xformVector(f.edgeU), xformVector(f.edgeV), twoSided,
                               materialIndex, xformVector(f.normal));
    }
}

uint32_t SceneResources::addMesh(const MeshVertex* vertices, uint32_t vertexCount, const uint32_t* indices,
                                 uint32_t indexCount, const float4x4& localToWorld, uint32_t materialIndex,
                                 std::string name) {
    if (!vertices || vertexCount == 0 || !indices || indexCount == 0 || indexCount % 3 != 0) {
        return std::numeric_limits<uint32_t>::max();
    }
    Mesh mesh;
    mesh.vertices.assign(vertices, vertices + vertexCount);
    mesh.indices.assign(indices, indices + indexCount);
    mesh.localToWorld = localToWorld;
    mesh.materialIndex = materialIndex;
    mesh.name = name.empty() ? ("Mesh " + std::to_string(m_meshes.size())) : std::move(name);
    m_meshes.push_back(std::move(mesh));
    return static_cast<uint32_t>(m_meshes.size() - 1);
}

bool SceneResources::setMeshTransform(uint32_t meshIndex, const float4x4& localToWorld) {
    if (meshIndex >= m_meshes.size()) {
        return false;
    }
    m_meshes[meshIndex].localToWorld = localToWorld;
    return true;
}

void SceneResources::storeRectangleOriented(const float3& corner, const float3& edgeU, const float3& edgeV,
                                            bool twoSided, uint32_t materialIndex, const float3& desiredNormal) {
    if (m_rectangles.size() >= kMaxRectangles) {
        return;
    }
    constexpr float kTiny = std::numeric_limits<float>::min();
    const float uLenSq = dot(edgeU, edgeU);
    const float vLenSq = dot(edgeV, edgeV);
    if (uLenSq <= kTiny || vLenSq <= kTiny) {
        return;
    }
    const float3 n = cross(edgeU, edgeV);
    const float nLenSq = dot(n, n);
    if (nLenSq <= kTiny) {
        return;
    }
    float3 unitNormal = n / std::sqrt(nLenSq);
    float3 target = unitNormal;
    const float desiredLenSq = dot(desiredNormal, desiredNormal);
    if (desiredLenSq > kTiny) {
        target = desiredNormal / std::sqrt(desiredLenSq);
    }
    if (dot(unitNormal, target) < 0.0f) {
        unitNormal = -unitNormal;
    }
    if (!std::isfinite(unitNormal.x) || !std::isfinite(unitNormal.y) || !std::isfinite(unitNormal.z)) {
        return;
    }
    PtrRect r;
    std::memset(&r, 0, sizeof(r));
    set4(r.corner, corner, 0.0f);
    set4(r.edgeU, edgeU, 1.0f / uLenSq);
    set4(r.edgeV, edgeV, 1.0f / vLenSq);
    set4(r.normalAndPlane, unitNormal, dot(unitNormal, corner));
    r.materialTwoSided[0] = materialIndex;
    r.materialTwoSided[1] = twoSided ? 1u : 0u;
    m_rectangles.push_back(r);
}

bool SceneResources::loadEnvironmentMap(const std::string& path, std::string& error) {
    std::vector<float> rgba;
    uint32_t w = 0, h = 0;
    if (!LoadHdrImageRgba(path, rgba, w, h, error)) {
        return false;
    }
    setEnvironmentPixels(std::move(rgba), w, h);
    return true;
}

void SceneResources::setEnvironmentPixels(std::vector<float> rgba, uint32_t width, uint32_t height) {
    m_envRgba = std::move(rgba);
    m_envWidth = width;
    m_envHeight = height;
}

uint32_t SceneResources::addTexture(const uint8_t* rgba8, uint32_t width, uint32_t height, bool srgb, uint32_t wrapS, uint32_t wrapT,
                                    uint32_t filter) {
    Texture t;
    t.width = width;
    t.height = height;
    t.wrapS = std::min(wrapS, 2u);
    t.wrapT = std::min(wrapT, 2u);
    t.filter = filter ? 1u : 0u;
    // 8-bit -> float through a 256-entry table; the sRGB transfer function is the piecewise IEC 61966-2-1 one
    float linear[256], plain[256];
    for (int i = 0; i < 256; ++i) {
        const float c = static_cast<float>(i) / 255.0f;
        plain[i] = c;
        linear[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
    }
    const size_t n = static_cast<size_t>(width) * height;
    t.rgba.resize(n * 4u);
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) t.rgba[i * 4 + c] = (srgb ? linear : plain)[rgba8[i * 4 + c]];
        t.rgba[i * 4 + 3] = plain[rgba8[i * 4 + 3]];
    }
    m_textures.push_back(std::move(t));
    return static_cast<uint32_t>(m_textures.size() - 1u);
}

void SceneResources::fillSceneDesc(PtrSceneDesc& desc) const {
    std::memset(&desc, 0, sizeof(desc));
    desc.spheres = m_spheres.empty() ? nullptr : m_spheres.data();
    desc.sphereCount = sphereCount();
    desc.rects = m_rectangles.empty() ? nullptr : m_rectangles.data();
    desc.rectCount = rectangleCount();
    desc.materials = m_materials.empty() ? nullptr : m_materials.data();
    desc.materialCount = materialCount();

    m_meshDescs.clear();
    m_meshPositions.assign(m_meshes.size(), {});
    m_meshNormals.assign(m_meshes.size(), {});
    m_meshUv0.assign(m_meshes.size(), {});
    m_meshUv1.assign(m_meshes.size(), {});
    m_meshTangents.assign(m_meshes.size(), {});
    for (size_t i = 0; i < m_meshes.size(); ++i) {
        const Mesh& mesh = m_meshes[i];
        auto& pos = m_meshPositions[i];
        auto& nrm = m_meshNormals[i];
        pos.resize(mesh.vertices.size() * 3);
        nrm.resize(mesh.vertices.size() * 3);
        for (size_t v = 0; v < mesh.vertices.size(); ++v) {
            pos[v * 3 + 0] = mesh.vertices[v].position.x;
            pos[v * 3 + 1] = mesh.vertices[v].position.y;
            pos[v * 3 + 2] = mesh.vertices[v].position.z;
            nrm[v * 3 + 0] = mesh.vertices[v].normal.x;
            nrm[v * 3 + 1] = mesh.vertices[v].normal.y;
            nrm[v * 3 + 2] = mesh.vertices[v].normal.z;
        }
        PtrMeshDesc d;
        std::memset(&d, 0, sizeof(d));
        d.positions = pos.data();
        d.normals = nrm.data();
        d.indices = mesh.indices.data();
        d.vertexCount = static_cast<uint32_t>(mesh.vertices.size());
        d.indexCount = static_cast<uint32_t>(mesh.indices.size());
        for (int c = 0; c < 4; ++c) {
            d.localToWorld[c * 4 + 0] = mesh.localToWorld.columns[c].x;
            d.localToWorld[c * 4 + 1] = mesh.localToWorld.columns[c].y;
            d.localToWorld[c * 4 + 2] = mesh.localToWorld.columns[c].z;
            d.localToWorld[c * 4 + 3] = mesh.localToWorld.columns[c].w;
        }
        d.materialIndex = mesh.materialIndex;
        if (mesh.hasUv0 || mesh.hasUv1) {
            auto& a = m_meshUv0[i];
            auto& b = m_meshUv1[i];
            a.resize(mesh.vertices.size() * 2);
            b.resize(mesh.vertices.size() * 2);
            for (size_t v = 0; v < mesh.vertices.size(); ++v) {
                a[v * 2] = mesh.vertices[v].uv.x;
                a[v * 2 + 1] = mesh.vertices[v].uv.y;
                b[v * 2] = mesh.vertices[v].uv1.x;
                b[v * 2 + 1] = mesh.vertices[v].uv1.y;
            }
            if (mesh.hasUv0) d.uv0 = a.data();
            if (mesh.hasUv1) d.uv1 = b.data();
        }
        if (mesh.hasTangents) {
            auto& t = m_meshTangents[i];
            t.resize(mesh.vertices.size() * 4);
            for (size_t v = 0; v < mesh.vertices.size(); ++v) {
                t[v * 4] = mesh.vertices[v].tangent.x;
                t[v * 4 + 1] = mesh.vertices[v].tangent.y;
                t[v * 4 + 2] = mesh.vertices[v].tangent.z;
                t[v * 4 + 3] = mesh.vertices[v].tangent.w;
            }
            d.tangents = t.data();
        }
        m_meshDescs.push_back(d);
    }
    desc.meshes = m_meshDescs.empty() ? nullptr : m_meshDescs.data();
    desc.meshCount = static_cast<uint32_t>(m_meshDescs.size());
    m_textureDescs.clear();
    for (const Texture& t : m_textures) {
        PtrTexture d;
        std::memset(&d, 0, sizeof(d));
        d.rgba = t.rgba.data();
        d.width = t.width;
        d.height = t.height;
        d.wrapS = t.wrapS;
        d.wrapT = t.wrapT;
        d.filter = t.filter;
        m_textureDescs.push_back(d);
    }
    desc.textures = m_textureDescs.empty() ? nullptr : m_textureDescs.data();
    desc.textureCount = static_cast<uint32_t>(m_textureDescs.size());
    if (!m_envRgba.empty() && m_envWidth > 0 && m_envHeight > 0) {
        desc.envRgba = m_envRgba.data();
        desc.envWidth = m_envWidth;
        desc.envHeight = m_envHeight;
    }
}

}  // namespace ptr
