// Material textures of the metallic-roughness model (PTR_METAL_PBR): storage, filtering, and the per-hit material the
// reference's Metal kernel builds from them (shaders/pathtrace.metal:5919-6400, helpers 108-198, 583-940, 2923-3216).
//
// The reference samples through the GPU's texture units (trilinear / anisotropic, `gradient2d` at the first hit from Igehy ray
// differentials, ray cones afterwards).  A software path has to fix its own filtering rule, and this is it - the same rule the
// oracle restates:
//   * storage: linear RGBA32F (sRGB decoded at load), every texture with its full mip chain (2x2 box filter, odd sizes clamp the
//     second tap), all levels of all textures in ONE float4 array in HBM;
//   * level of detail: ray cones only (ray_cone_lod_from_footprint, :162-176): log2(footprint x uvPerWorld x max(W, H));
//   * filtering: bilinear inside a level (texel centres at (i + 0.5) / W, wrap per the glTF sampler), linear between the two
//     nearest levels; NEAREST samplers read one texel of the rounded level.
// What the reference derives from ray differentials at the first hit (anisotropic sampling, the normal-variance term of the
// roughness widening) has no counterpart here.
#pragma once

#include "device_types.h"
#include "vec.h"

namespace ptrk {

constexpr uint32_t kNoTexture = 0xFFFFFFFFu;

// per-texture record in HBM: 5 uint4
//   [0] = (width, height, mipCount, wrapS | wrapT << 2 | filter << 4)   [1..4] = texel offset of levels 0..15 in the texel array
constexpr uint32_t kTexInfoVec4 = 5u;

struct TexLevel {
    uint32_t offset, width, height;
};

__device__ __forceinline__ TexLevel texLevel(const SceneView& sc, uint32_t tex, uint32_t level, uint4 head) {
    const uint4 offs = sc.texInfo[tex * kTexInfoVec4 + 1u + (level >> 2)];
    const uint32_t l = level & 3u;
    TexLevel r;
    r.offset = l == 0u ? offs.x : (l == 1u ? offs.y : (l == 2u ? offs.z : offs.w));
    r.width = max(head.x >> level, 1u);
    r.height = max(head.y >> level, 1u);
    return r;
}

__device__ __forceinline__ int texWrap(int i, int n, uint32_t mode) {
    if (mode == 1u) return min(max(i, 0), n - 1);
    if (mode == 2u) {
        const int period = 2 * n;
        int j = i % period;
        if (j < 0) j += period;
        return j < n ? j : period - 1 - j;
    }
    int j = i % n;
    return j < 0 ? j + n : j;
}

__device__ __forceinline__ float4 texBilinear(const SceneView& sc, const TexLevel& L, float u, float v, uint32_t flags) {
    const uint32_t wrapS = flags & 3u, wrapT = (flags >> 2) & 3u;
    const int W = static_cast<int>(L.width), H = static_cast<int>(L.height);
    if (((flags >> 4) & 1u) == 0u) {   // NEAREST
        const int x = texWrap(static_cast<int>(floorf(u * static_cast<float>(W))), W, wrapS);
        const int y = texWrap(static_cast<int>(floorf(v * static_cast<float>(H))), H, wrapT);
        return sc.texels[L.offset + static_cast<uint32_t>(y) * L.width + static_cast<uint32_t>(x)];
    }
    const float fx = u * static_cast<float>(W) - 0.5f, fy = v * static_cast<float>(H) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = texWrap(static_cast<int>(x0f), W, wrapS), x1 = texWrap(static_cast<int>(x0f) + 1, W, wrapS);
    const int y0 = texWrap(static_cast<int>(y0f), H, wrapT), y1 = texWrap(static_cast<int>(y0f) + 1, H, wrapT);
    const float4* base = sc.texels + L.offset;
    const float4 c00 = base[static_cast<uint32_t>(y0) * L.width + static_cast<uint32_t>(x0)];
    const float4 c10 = base[static_cast<uint32_t>(y0) * L.width + static_cast<uint32_t>(x1)];
    const float4 c01 = base[static_cast<uint32_t>(y1) * L.width + static_cast<uint32_t>(x0)];
    const float4 c11 = base[static_cast<uint32_t>(y1) * L.width + static_cast<uint32_t>(x1)];
    const float ix = 1.0f - tx, iy = 1.0f - ty;
    return make_float4((c00.x * ix + c10.x * tx) * iy + (c01.x * ix + c11.x * tx) * ty, (c00.y * ix + c10.y * tx) * iy + (c01.y * ix + c11.y * tx) * ty,
                       (c00.z * ix + c10.z * tx) * iy + (c01.z * ix + c11.z * tx) * ty, (c00.w * ix + c10.w * tx) * iy + (c01.w * ix + c11.w * tx) * ty);
}

// sample_material_texture_level with the filtering rule above; `fallback` when the slot has no texture
__device__ __forceinline__ float4 texSample(const SceneView& sc, uint32_t tex, float u, float v, float lod, float4 fallback) {
    if (tex == kNoTexture || tex >= sc.textureCount) return fallback;
    const uint4 head = sc.texInfo[tex * kTexInfoVec4];
    const float maxMip = static_cast<float>(head.z - 1u);
    const float l = fminf(fmaxf(lod, 0.0f), maxMip);
    if (((head.w >> 4) & 1u) == 0u) return texBilinear(sc, texLevel(sc, tex, static_cast<uint32_t>(floorf(l + 0.5f)), head), u, v, head.w);
    const float l0f = floorf(l);
    const uint32_t l0 = static_cast<uint32_t>(l0f), l1 = min(l0 + 1u, head.z - 1u);
    const float f = l - l0f;
    const float4 a = texBilinear(sc, texLevel(sc, tex, l0, head), u, v, head.w);
    if (!(f > 0.0f) || l1 == l0) return a;
    const float4 b = texBilinear(sc, texLevel(sc, tex, l1, head), u, v, head.w);
    return make_float4(a.x + (b.x - a.x) * f, a.y + (b.y - a.y) * f, a.z + (b.z - a.z) * f, a.w + (b.w - a.w) * f);
}

// ray_cone_lod_from_footprint (:162-176)
__device__ __forceinline__ float texLod(const SceneView& sc, uint32_t tex, float uvPerWorld, float footprintWorld) {
    if (tex == kNoTexture || tex >= sc.textureCount) return 0.0f;
    const uint4 head = sc.texInfo[tex * kTexInfoVec4];
    if (head.x == 0u || head.y == 0u) return 0.0f;
    if (head.z <= 1u || uvPerWorld <= 0.0f || footprintWorld <= 0.0f) return 0.0f;
    const float maxRes = fmaxf(static_cast<float>(head.x), static_cast<float>(head.y));
    const float texelFootprint = footprintWorld * uvPerWorld * maxRes;
    const float lod = log2f(fmaxf(texelFootprint, 1.0e-7f));
    return fminf(fmaxf(lod, 0.0f), static_cast<float>(head.z - 1u));
}

// decode_normal_map (:108-127), flipGreen = false
__device__ __forceinline__ f3 decodeNormalMap(f3 s, float normalScale, float& outLength) {
    f3 n = s * 2.0f - mk3(1.0f);
    n.x *= normalScale;
    n.y *= normalScale;
    outLength = length(n);
    const float xyLen2 = n.x * n.x + n.y * n.y;
    n.z = sqrtf(smax(1.0f - xyLen2, 0.0f));
    const float len2 = dot(n, n);
    if (len2 > 1.0e-12f) {
        n = n * (1.0f / sqrtf(len2));
    } else {
        n = mk3(0.0f, 0.0f, 1.0f);
    }
    return n;
}

// One texture slot's sampling context (make_pbr_texture_sampling_context, :3018-3056, without the first-hit gradients):
// transformed coordinates of the slot's uv set and its uv-per-world scale.
struct TexSlot {
    float u, v, uvPerWorld;
};

__device__ __forceinline__ TexSlot texSlot(const float4* mraw, uint32_t slot, uint32_t uvSet, float2 uv0, float2 uv1, float perWorld0, float perWorld1) {
    // rows of KHR_texture_transform live behind the compact material record: raw[2*slot], raw[2*slot + 1] (xyz)
    f3 row0 = mk3(mraw[2u * slot]), row1 = mk3(mraw[2u * slot + 1u]);
    const float linearSum = (fabsf(row0.x) + fabsf(row0.y)) + (fabsf(row1.x) + fabsf(row1.y));
    if (!finite3(row0) || !finite3(row1) || !(linearSum > 1.0e-8f)) {   // pbr_texture_transform_rows (:2942-2983)
        row0 = mk3(1.0f, 0.0f, 0.0f);
        row1 = mk3(0.0f, 1.0f, 0.0f);
    }
    const float2 uv = uvSet == 0u ? uv0 : uv1;
    const float perWorld = uvSet == 0u ? perWorld0 : perWorld1;
    TexSlot t;
    t.u = (row0.x * uv.x + row0.y * uv.y) + row0.z;
    t.v = (row1.x * uv.x + row1.y * uv.y) + row1.z;
    const float sx = sqrtf(row0.x * row0.x + row1.x * row1.x), sy = sqrtf(row0.y * row0.y + row1.y * row1.y);
    t.uvPerWorld = perWorld * smax(smax(sx, sy), 1.0e-6f);   // pbr_transform_uv_per_world (:3002-3009)
    return t;
}

}  // namespace ptrk
