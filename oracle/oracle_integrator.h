// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md.
//
// CPU restatement of the reference's Embree-path integrator,
// src/headless/EmbreeHeadlessRenderer.mm (camera 150-232, background/environment 234-310, BSDF helpers
// 312-885, lights 887-1168, car paint 1170-1313, EvaluateBsdf 1315-1491, SampleBsdf 1493-1918,
// render loop 2443-3214) and of src/renderer/EnvImportanceSampler.mm (alias tables 16-171, sampling 173-236).
//
// PARITY PIN STATUS: "parity unpinned" at content level.  The reference publishes a single content pin for
// this path (paper/paper.md:183-188: 66,925-byte EXR with a sha256 produced on Apple Silicon by the real
// Embree + Apple libm build).  The byte count is reproduced exactly by tests/test_oracle.py; the sha256 is
// not reproducible off that platform (different libm, FMA contraction and Embree ISA), and the reference's
// Objective-C++/Embree sources cannot be built in this image.  Everything else the reference holds for this
// path (smoke script, EmbreeSmokeTest hit/no-hit) is existence-level and is covered.
#pragma once

#include <cstdint>
#include <vector>

#include "oracle_math.h"
#include "oracle_raycast.h"
#include "ptr_abi.h"

namespace oracle {

struct Rng {  // EmbreeHeadlessRenderer.mm:52-68
    uint32_t state = 1u;
    static uint32_t hash(uint32_t x) {
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        return x;
    }
    float nextFloat() {
        state = hash(state);
        return static_cast<float>(state & 0x00FFFFFFu) / 16777216.0f;
    }
};

struct Camera {  // CameraBasis, :37-45
    V3 origin, lowerLeft, horizontal, vertical, u, v;
    float lensRadius = 0.0f;
};

struct Ray {
    V3 origin, direction;
};

struct EnvDistribution {  // EnvImportanceSampler.h:12-22
    std::vector<float> texelPdf;
    std::vector<uint32_t> conditionalAlias;
    std::vector<float> conditionalThreshold;
    std::vector<uint32_t> marginalAlias;
    std::vector<float> marginalThreshold;
    uint32_t width = 0, height = 0, aliasCount = 0;
    float totalWeight = 0.0f;
};

struct EnvMap {
    const float* rgba = nullptr;
    uint32_t width = 0, height = 0;
    EnvDistribution dist;
    bool hasDistribution = false;
};

struct EnvSample {
    V3 direction, radiance;
    float pdf = 0.0f;
};

struct ClampParams {  // FireflyClampParams, :136-144
    float clampFactor = 0.0f, clampFloor = 0.0f, throughputClamp = 0.0f;
    float specularTailClampBase = 0.0f, specularTailClampRoughnessScale = 0.0f;
    float minSpecularPdf = 1.0e-8f, enabled = 0.0f;
    bool thinDielectrics = false;   // PTR_METAL_THIN (Metal-only semantics, shaders/pathtrace.metal:4589-4592, 5649-5659)
    bool metalSpecular = false;     // PTR_METAL_SPECULAR: VNDF sampling, G1 pdf, energy compensation for rough metals
    bool metalSss = false;          // PTR_METAL_SSS: type 5 evaluates to zero; separable diffusion sampling when sssMode == 1,
    uint32_t sssMode = 0;           // random walk when sssMode == 2 on materials that ask for it
    uint32_t sssMaxSteps = 32;
    bool metalPbr = false;          // PTR_METAL_PBR: three-lobe metallic-roughness model of the Metal integrator
    bool metalClamps = false;       // PTR_METAL_CLAMPS: Metal variants of the clamps (shaders/pathtrace.metal:3550-3633, Appendix A rows 4-6)
    float maxContribution = 0.0f;   // fireflyClampMaxContribution
    float minSpecularPdfRaw = 0.0f;
};

struct BsdfEval {
    V3 value;
    float pdf = 0.0f;
    bool isDelta = false;
};

struct BsdfSample {
    V3 direction, weight;
    float pdf = 0.0f;
    bool isDelta = false;
    int mediumEvent = 0;   // Metal-only: +1 refracted in through a front face, -1 out through a back face (pathtrace.metal:5683)
    bool hasExitPoint = false;   // Metal-only subsurface samples: the path leaves at exitPoint
    V3 exitPoint;
    V3 exitNormal;               // random walk: outward normal at the exit (zero: use the shading normal of the entry)
    int lobe = 0;                // Metal metallic-roughness sampler: 0 diffuse, 1 specular, 2 transmission (ray-cone widening)
    float lobeRoughness = 0.0f;
};

struct HitInfo {  // :97-107
    V3 position, normal{0.0f, 1.0f, 0.0f}, shadingNormal{0.0f, 1.0f, 0.0f};
    float t = 0.0f;
    uint32_t materialIndex = 0;
    bool frontFace = true, twoSided = false;
    GeomType primitiveType = GeomType::Mesh;
    uint32_t primitiveIndex = 0;
    uint32_t geomIndex = 0;   // mesh index of a mesh hit (path signatures only)
    uint32_t geom = 0;        // Scene::geoms index of the hit geometry, barycentrics of a triangle hit (textured materials)
    float bu = 0.0f, bv = 0.0f;
};

struct RenderCounters {
    uint64_t extendRays = 0, shadowRays = 0, nodes = 0, prims = 0, shadedHits = 0, triangleHits = 0;
};

Camera buildCamera(const PtrSettings& s);
Ray generateCameraRay(const Camera& cam, uint32_t width, uint32_t height, uint32_t x, uint32_t y, Rng& rng);
ClampParams makeClampParams(const PtrSettings& s);
bool buildEnvDistribution(const float* rgba, uint32_t width, uint32_t height, EnvDistribution& out);
EnvSample sampleEnvironmentCpu(const EnvDistribution& dist, float uMarginal, float uConditional, float uJitter,
                               float rotation, float intensity, const float* rgba);
V3 sampleEnvironment(const EnvMap& env, V3 direction, float rotation, float intensity);
float environmentPdf(const EnvMap& env, float rotation, V3 direction);
BsdfEval evaluateBsdf(const PtrMaterial& m, V3 position, V3 normal, V3 wo, V3 wi, const ClampParams& cp);
BsdfSample sampleBsdf(const PtrMaterial& m, V3 position, V3 normal, V3 wo, V3 incidentDir, bool frontFace, Rng& rng,
                      const ClampParams& cp);
bool intersectScene(const Scene& scene, const Ray& ray, HitInfo& out, Counters* counters = nullptr);
V3 nextRayOrigin(const HitInfo& hit, V3 direction);   // OffsetRayOrigin, E:917-931 (for the function-level tests)
// Samples texture `texture` of the scene with the filtering rule of the textured metallic-roughness model: in n * {u, v, lod}, out n * RGBA.
void sampleTextures(const PtrSceneDesc& desc, uint32_t texture, const float* in, uint64_t n, float* out);

// Renders pixels with y in [y0, y1) (full width); out_rgb is the full W*H*3 image, other rows untouched.
// out_signature / out_marginal (optional, W*H each): path signature of every pixel's last sample and whether one of its
// rectangle-light shadow tests is decided within rounding noise of the ray length (deterministic-stream diagnostics).
void render(const Scene& scene, const PtrSceneDesc& desc, const PtrSettings& settings, uint32_t spp,
            uint32_t threads, uint32_t y0, uint32_t y1, float* out_rgb, RenderCounters* counters,
            uint32_t* out_signature = nullptr, uint8_t* out_marginal = nullptr);

}  // namespace oracle
