// Environment-map importance distribution: luminance x solid-angle weights, Vose alias tables for the
// marginal (rows) and per-row conditional (columns) choices, and the per-texel solid-angle pdf.
// Behaviour follows the reference's src/renderer/EnvImportanceSampler.mm:16-171 (interface
// include/renderer/EnvImportanceSampler.h:12-33); tables are emitted in the packed (threshold, alias)
// pairs the device sampler reads (reference: EnvironmentAliasEntry, include/MetalShaderTypes.h:99-104).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ptr {

struct AliasEntry {
    float threshold = 1.0f;
    uint32_t alias = 0;
};

struct EnvImportanceDistribution {
    std::vector<float> texelPdf;            // width*height, solid-angle domain
    std::vector<AliasEntry> conditional;    // width*height
    std::vector<AliasEntry> marginal;       // height
    uint32_t width = 0;
    uint32_t height = 0;
    float totalWeight = 0.0f;
};

bool BuildEnvImportanceDistribution(const float* rgba32, uint32_t width, uint32_t height,
                                    EnvImportanceDistribution* outDist, std::string* error = nullptr);

}  // namespace ptr
