#include "env_importance_sampler.h"

#include <algorithm>
#include <cmath>

namespace ptr {
namespace {

constexpr float kPi = 3.14159265358979323846f;

// Vose's alias method.  Work lists are consumed from the back, and an over-full entry moves to the
// under-full list once it drops below 1 - 1e-7, exactly like the reference (the tables, not just the
// distribution they encode, feed the random stream -> texel mapping).
void buildAlias(const float* prob, size_t count, AliasEntry* out) {
    std::vector<float> scaled(count);
    std::vector<uint32_t> under, over;
    under.reserve(count);
    over.reserve(count);
    for (size_t i = 0; i < count; ++i) {
        scaled[i] = prob[i] * static_cast<float>(count);
        out[i] = AliasEntry{0.0f, 0u};
        if (scaled[i] < 1.0f) under.push_back(static_cast<uint32_t>(i));
        else over.push_back(static_cast<uint32_t>(i));
    }
    while (!under.empty() && !over.empty()) {
        const uint32_t u = under.back();
        under.pop_back();
        const uint32_t o = over.back();
        out[u].threshold = std::min(std::max(scaled[u], 0.0f), 1.0f);
        out[u].alias = o;
        scaled[o] = (scaled[o] + scaled[u]) - 1.0f;
        if (scaled[o] < 1.0f - 1e-7f) {
            over.pop_back();
            under.push_back(o);
        }
    }
    for (uint32_t i : under) out[i] = AliasEntry{1.0f, i};
    for (uint32_t i : over) out[i] = AliasEntry{1.0f, i};
}

}  // namespace

bool BuildEnvImportanceDistribution(const float* rgba, uint32_t width, uint32_t height,
                                    EnvImportanceDistribution* dist, std::string* error) {
    if (!dist) {
        if (error) *error = "Output distribution pointer was null";
        return false;
    }
    *dist = EnvImportanceDistribution{};
    if (!rgba || width == 0 || height == 0) {
        if (error) *error = "Invalid environment texture data";
        return false;
    }
    const size_t texels = static_cast<size_t>(width) * height;
    const float dTheta = kPi / static_cast<float>(height);
    const float dPhi = (2.0f * kPi) / static_cast<float>(width);

    std::vector<float> weight(texels), rowWeight(height, 0.0f), cell(height);
    float total = 0.0f;
    for (uint32_t y = 0; y < height; ++y) {
        cell[y] = std::max(std::sin((static_cast<float>(y) + 0.5f) * dTheta), 0.0f) * dTheta * dPhi;
        float* w = weight.data() + static_cast<size_t>(y) * width;
        const float* px = rgba + static_cast<size_t>(y) * width * 4u;
        for (uint32_t x = 0; x < width; ++x, px += 4) {
            const float lum = (0.2126f * px[0] + 0.7152f * px[1]) + 0.0722f * px[2];
            w[x] = std::max(lum, 0.0f) * cell[y];
            rowWeight[y] += w[x];
            total += w[x];   // running float sum in scan order (matters for bit parity of the pdfs)
        }
    }
    if (total <= 0.0f) {
        if (error) *error = "Environment map contains no positive radiance";
        return false;
    }
    dist->width = width;
    dist->height = height;
    dist->totalWeight = total;

    std::vector<float> prob(std::max(width, height));
    for (uint32_t y = 0; y < height; ++y) prob[y] = rowWeight[y] > 0.0f ? rowWeight[y] / total : 0.0f;
    dist->marginal.resize(height);
    buildAlias(prob.data(), height, dist->marginal.data());

    dist->conditional.resize(texels);
    dist->texelPdf.resize(texels);
    for (uint32_t y = 0; y < height; ++y) {
        const float* w = weight.data() + static_cast<size_t>(y) * width;
        if (rowWeight[y] > 0.0f) {
            const float inv = 1.0f / rowWeight[y];
            for (uint32_t x = 0; x < width; ++x) prob[x] = w[x] * inv;
        } else {
            std::fill(prob.begin(), prob.begin() + width, 1.0f / static_cast<float>(width));
        }
        buildAlias(prob.data(), width, dist->conditional.data() + static_cast<size_t>(y) * width);
        float* pdf = dist->texelPdf.data() + static_cast<size_t>(y) * width;
        for (uint32_t x = 0; x < width; ++x) pdf[x] = cell[y] > 0.0f ? (w[x] / total) / cell[y] : 0.0f;
    }
    return true;
}

}  // namespace ptr
