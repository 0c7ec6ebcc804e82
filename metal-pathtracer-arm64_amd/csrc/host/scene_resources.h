// SceneResources — CPU scene arrays handed to a headless backend.
// Behavioural mirror of the CPU half of the reference's SceneResources
// (src/renderer/SceneResources.mm: addMaterial 902-1037, addSphere 1441-1455,
// addRectangle 1743-1833, addBoxTransformed 1843-1895, storeRectangleOriented 2317-2365);
// the Metal buffer/texture half has no counterpart here (device upload lives in hip_scene.cpp).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "ptr_abi.h"
#include "vecmath.h"

namespace ptr {

constexpr uint32_t kMaxSpheres = 512;     // include/MetalShaderTypes.h:15-17
constexpr uint32_t kMaxMaterials = 512;
constexpr uint32_t kMaxRectangles = 128;
constexpr uint32_t kInvalidTextureIndex = 0xFFFFFFFFu;

enum class MaterialType : uint32_t {
    Lambertian = 0,
    Metal = 1,
    Dielectric = 2,
    DiffuseLight = 3,
    Plastic = 4,
    Subsurface = 5,
    CarPaint = 6,
    PbrMetallicRoughness = 7,
};

// Named-argument form of the reference's 35-parameter addMaterial overload.
struct MaterialParams {
    float3 baseColor{1.0f, 1.0f, 1.0f};
    float roughness = 0.0f;
    MaterialType type = MaterialType::Lambertian;
    float indexOfRefraction = 1.5f;
    float3 emission{0.0f, 0.0f, 0.0f};
    bool emissionUsesEnvironment = false;
    float3 conductorEta{0.0f, 0.0f, 0.0f};
    float3 conductorK{0.0f, 0.0f, 0.0f};
    bool hasConductorParameters = false;
    float coatRoughness = 0.0f;
    float coatThickness = 0.0f;
    float3 coatTint{1.0f, 1.0f, 1.0f};
    float3 coatAbsorption{0.0f, 0.0f, 0.0f};
    float coatIor = 1.5f;
    float3 dielectricSigmaA{0.0f, 0.0f, 0.0f};
    float3 sssSigmaA{0.0f, 0.0f, 0.0f};
    float3 sssSigmaS{0.0f, 0.0f, 0.0f};
    float sssMeanFreePath = 0.0f;
    float sssAnisotropy = 0.0f;
    uint32_t sssMethod = 0;
    bool sssCoatEnabled = false;
    bool sssSigmaOverride = false;
    float carpaintBaseMetallic = 0.0f;
    float carpaintBaseRoughness = 0.0f;
    float carpaintFlakeSampleWeight = 0.0f;
    float carpaintFlakeRoughness = 0.0f;
    float carpaintFlakeAnisotropy = 0.0f;
    float carpaintFlakeNormalStrength = 0.0f;
    float carpaintFlakeScale = 1.0f;
    float carpaintFlakeReflectanceScale = 1.0f;
    float3 carpaintBaseEta{0.0f, 0.0f, 0.0f};
    float3 carpaintBaseK{0.0f, 0.0f, 0.0f};
    bool carpaintHasBaseConductor = false;
    float3 carpaintBaseTint{1.0f, 1.0f, 1.0f};
    bool thinDielectric = false;
    std::string name;
};

class SceneResources {
public:
    struct MeshVertex {  // include/renderer/SceneResources.h:145-151
        float3 position{0.0f, 0.0f, 0.0f};
        float3 normal{0.0f, 1.0f, 0.0f};
        float2 uv{};
        float2 uv1{};
        float4 tangent{0.0f, 0.0f, 0.0f, 0.0f};   // object space, w = handedness; w = 0: the mesh has no tangents
    };

    // A material texture as the device and the oracle consume it: level 0, linear RGBA floats, with its sampler state
    // (the CPU half of SceneResources::materialTextures / MaterialTextureInfo, src/renderer/SceneResources.mm:213-420).
    struct Texture {
        std::vector<float> rgba;
        uint32_t width = 0, height = 0;
        uint32_t wrapS = 0, wrapT = 0;   // 0 repeat, 1 clamp to edge, 2 mirrored repeat
        uint32_t filter = 1;             // 0 nearest, 1 linear
    };

    struct Mesh {
        std::vector<MeshVertex> vertices;
        std::vector<uint32_t> indices;
        float4x4 localToWorld = float4x4::identity();
        uint32_t materialIndex = 0;
        std::string name;
        bool hasUv0 = false, hasUv1 = false, hasTangents = false;
    };

    void clear();

    // 8-bit RGBA pixels (row 0 = top) -> linear float texture; srgb: the colour channels are sRGB-encoded (base colour, emissive)
    uint32_t addTexture(const uint8_t* rgba8, uint32_t width, uint32_t height, bool srgb, uint32_t wrapS, uint32_t wrapT, uint32_t filter);
    const std::vector<Texture>& textures() const { return m_textures; }
    Mesh& meshAt(uint32_t index) { return m_meshes[index]; }

    uint32_t addMaterial(const MaterialParams& params);
    // Raw MaterialData (glTF path: BuildGltfMaterial -> addMaterialData, SceneResources.mm:1088-1102)
    uint32_t addMaterialData(const PtrMaterial& material, std::string name = {});
    void addSphere(const float3& center, float radius, uint32_t materialIndex);
    void addRectangle(const float3& boundsMin, const float3& boundsMax, uint32_t normalAxis,
                      bool normalPositive, bool twoSided, uint32_t materialIndex);
    void addBox(const float3& minCorner, const float3& maxCorner, uint32_t materialIndex,
                bool includeBottomFace = true, bool twoSided = false);
    void addBoxTransformed(const float3& minCorner, const float3& maxCorner, uint32_t materialIndex,
                           const float4x4& transform, bool includeBottomFace = true, bool twoSided = false);
    uint32_t addMesh(const MeshVertex* vertices, uint32_t vertexCount, const uint32_t* indices,
                     uint32_t indexCount, const float4x4& localToWorld, uint32_t materialIndex,
                     std::string name = {});

    const std::vector<Mesh>& meshes() const { return m_meshes; }
    const float4x4& meshTransform(uint32_t meshIndex) const { return m_meshes[meshIndex].localToWorld; }
    bool setMeshTransform(uint32_t meshIndex, const float4x4& localToWorld);

    const PtrSphere* spheresData() const { return m_spheres.data(); }
    uint32_t sphereCount() const { return static_cast<uint32_t>(m_spheres.size()); }
    const PtrRect* rectanglesData() const { return m_rectangles.data(); }
    uint32_t rectangleCount() const { return static_cast<uint32_t>(m_rectangles.size()); }
    const PtrMaterial* materialsData() const { return m_materials.data(); }
    uint32_t materialCount() const { return static_cast<uint32_t>(m_materials.size()); }
    const std::string& materialName(uint32_t index) const { return m_materialNames[index]; }

    // Environment map decoded to linear RGBA32F (what the Embree backend's LoadEnvironmentMap yields,
    // EmbreeHeadlessRenderer.mm:1920-2021).
    bool loadEnvironmentMap(const std::string& path, std::string& error);
    void setEnvironmentPixels(std::vector<float> rgba, uint32_t width, uint32_t height);
    const std::vector<float>& environmentRgba() const { return m_envRgba; }
    uint32_t environmentWidth() const { return m_envWidth; }
    uint32_t environmentHeight() const { return m_envHeight; }

    // Flat C-ABI view of the arrays (pointers valid while *this is unchanged).
    void fillSceneDesc(PtrSceneDesc& desc) const;

private:
    void storeRectangleOriented(const float3& corner, const float3& edgeU, const float3& edgeV,
                                bool twoSided, uint32_t materialIndex, const float3& desiredNormal);

    std::vector<PtrSphere> m_spheres;
    std::vector<PtrRect> m_rectangles;
    std::vector<PtrMaterial> m_materials;
    std::vector<std::string> m_materialNames;
    std::vector<Mesh> m_meshes;
    std::vector<Texture> m_textures;
    std::vector<float> m_envRgba;
    uint32_t m_envWidth = 0;
    uint32_t m_envHeight = 0;

    // scratch for fillSceneDesc
    mutable std::vector<PtrMeshDesc> m_meshDescs;
    mutable std::vector<std::vector<float>> m_meshPositions;
    mutable std::vector<std::vector<float>> m_meshNormals;
    mutable std::vector<std::vector<float>> m_meshUv0, m_meshUv1, m_meshTangents;
    mutable std::vector<PtrTexture> m_textureDescs;
};

}  // namespace ptr
