// RenderSettings — the knobs of the headless render path.
// Mirrors the fields/defaults of the reference's include/renderer/RenderSettings.h:16-145 that the
// headless path reads (GUI/bloom/denoise/debug toggles are parsed-and-ignored by SceneManager).
#pragma once

#include <cstdint>
#include <string>

#include "vecmath.h"

namespace ptr {

struct RenderSettings {
    enum class BackgroundMode : uint32_t { Gradient = 0, Solid = 1, Environment = 2 };
    enum class SssMode : uint32_t { Off = 0, Separable = 1, RandomWalk = 2 };

    uint32_t samplesPerFrame = 1;
    uint32_t maxDepth = 50;
    bool enableRussianRoulette = true;
    uint32_t fixedRngSeed = 0;
    uint32_t renderWidth = 0;
    uint32_t renderHeight = 0;
    bool enableSoftwareRayTracing = false;
    SssMode sssMode = SssMode::Off;
    uint32_t sssMaxSteps = 32;
    bool enableSpecularNee = true;
    bool enableMnee = false;
    // Not in the reference's RenderSettings: which integrator semantics the HIP backend follows.  0 = the Embree
    // backend's (the parity oracle); PTR_METAL_MEDIA | PTR_METAL_THIN | PTR_METAL_FACE_NORMAL add what only the Metal kernel does
    // (absorbing dielectric interiors, thin-walled glass, ray-facing glass normals).
    uint32_t metalSemantics = 0;
    bool enableMneeSecondary = true;

    uint32_t tonemapMode = 1;  // 1=Linear, 2=ACES, 3=Reinhard, 4=Hable
    uint32_t acesVariant = 0;
    float exposure = 0.0f;
    float reinhardWhitePoint = 1.5f;

    // glTF loader options carried through the `renderer gltf*` directives
    bool gltfViewerCompatibilityMode = false;
    bool gltfThinWalledFallback = true;
    float gltfEmissiveScale = 1.0f;

    float3 cameraTarget{0.0f, 0.0f, 0.0f};
    float cameraDistance = 13.490737f;
    float cameraYaw = 0.226799f;
    float cameraPitch = 0.149000f;
    float cameraVerticalFov = 20.0f;
    float cameraDefocusAngle = 0.0f;
    float cameraFocusDistance = 0.0f;

    BackgroundMode backgroundMode = BackgroundMode::Gradient;
    float3 backgroundColor{0.0f, 0.0f, 0.0f};
    std::string environmentMapPath{};
    float environmentRotation = 0.0f;
    float environmentIntensity = 1.0f;

    bool fireflyClampEnabled = true;
    float fireflyClampFactor = 32.0f;
    float fireflyClampFloor = 4.0f;
    float throughputClamp = 32.0f;
    float specularTailClampBase = 0.0f;
    float specularTailClampRoughnessScale = 0.0f;
    float minSpecularPdf = 0.0f;
    float fireflyClampMaxContribution = 1000.0f;
};

}  // namespace ptr
