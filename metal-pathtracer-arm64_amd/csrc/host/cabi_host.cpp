// Host-side (no GPU) entry points of the C-ABI: scene loading and image output.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <filesystem>
#include <memory>
#include <string>
#include <vector>

#include "env_importance_sampler.h"
#include "headless.h"
#include "image_decoders.h"
#include "image_writer.h"
#include "ptr_abi.h"
#include "ptr_debug.h"
#include "scene_geometry.h"
#include "bvh_layout.h"
#include "tangent_space.h"
#include "scene_manager.h"

struct PtrHostScene {
    ptr::SceneResources resources;
    ptr::RenderSettings settings;
    PtrSceneDesc desc{};
    PtrSettings podSettings{};
};

namespace {
void setErr(char* err, size_t cap, const std::string& msg) {
    if (err && cap > 0) {
        std::snprintf(err, cap, "%s", msg.c_str());
    }
}

// Nothing may unwind across the C boundary (loaders allocate, parse and start threads): every entry point runs its
// body through this guard and reports exceptions through the error string like any other failure.
template <typename Fn>
int guarded(char* err, size_t cap, Fn&& body) {
    try {
        return body();
    } catch (const std::exception& e) {
        setErr(err, cap, std::string("exception: ") + e.what());
    } catch (...) {
        setErr(err, cap, "unknown exception");
    }
    return 1;
}
}  // namespace

extern "C" {

int ptr_host_scene_load(const char* scene_path, const char* asset_dir, PtrHostScene** out, char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!scene_path || !out) {
            setErr(err, err_cap, "ptr_host_scene_load: null argument");
            return 1;
        }
        auto scene = std::make_unique<PtrHostScene>();
        ptr::SceneManager manager(asset_dir ? std::string(asset_dir) : std::string());
        std::string error;
        if (!manager.loadSceneFromPath(scene_path, scene->resources, scene->settings, &error)) {
            setErr(err, err_cap, error);
            return 1;
        }
        scene->resources.fillSceneDesc(scene->desc);
        ptr::FillPtrSettings(scene->settings, scene->podSettings);
        // unlike render(), keep 0 when the scene gives no size so callers can apply the CLI default (1280x720)
        if (scene->settings.renderWidth == 0) scene->podSettings.width = 0;
        if (scene->settings.renderHeight == 0) scene->podSettings.height = 0;
        *out = scene.release();
        return 0;
    });
}

void ptr_host_scene_free(PtrHostScene* scene) { delete scene; }

int ptr_host_scene_desc(const PtrHostScene* scene, PtrSceneDesc* out_desc, PtrSettings* out_settings) {
    if (!scene) return 1;
    if (out_desc) *out_desc = scene->desc;
    if (out_settings) *out_settings = scene->podSettings;
    return 0;
}

int ptr_host_write_image(const char* path, const char* format, const float* linear_rgb, uint32_t width,
                         uint32_t height, int rgba_exr, uint32_t tonemap_mode, uint32_t aces_variant,
                         float exposure, float reinhard_white, char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!path || !format || !linear_rgb || width == 0 || height == 0) {
            setErr(err, err_cap, "ptr_host_write_image: bad argument");
            return 1;
        }
        ptr::ImageFileFormat fmt;
        if (!ptr::ParseImageFileFormat(format, fmt)) {
            setErr(err, err_cap, std::string("Unknown format: ") + format);
            return 1;
        }
        std::string error;
        bool ok;
        if (fmt == ptr::ImageFileFormat::EXR && rgba_exr) {
            std::vector<float> rgba(static_cast<size_t>(width) * height * 4u, 1.0f);
            for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) {
                rgba[i * 4 + 0] = linear_rgb[i * 3 + 0];
                rgba[i * 4 + 1] = linear_rgb[i * 3 + 1];
                rgba[i * 4 + 2] = linear_rgb[i * 3 + 2];
            }
            ok = ptr::WriteExrRgba(path, rgba.data(), width, height, "Linear sRGB", &error);
        } else {
            ptr::TonemapSettings tm;
            tm.tonemapMode = tonemap_mode == 0 ? 1u : tonemap_mode;
            tm.acesVariant = aces_variant;
            tm.exposure = exposure;
            tm.reinhardWhitePoint = reinhard_white;
            ok = ptr::WriteImage(path, fmt, linear_rgb, width, height, tm, &error);
        }
        if (!ok) {
            setErr(err, err_cap, error);
            return 1;
        }
        return 0;
    });
}

int ptr_host_write_exr_multilayer(const char* path, const float* linear_rgb, uint32_t width, uint32_t height,
                                  const float* sample_counts, const char* colorspace, char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!path || !linear_rgb || width == 0 || height == 0) {
            setErr(err, err_cap, "ptr_host_write_exr_multilayer: bad argument");
            return 1;
        }
        const size_t n = static_cast<size_t>(width) * height;
        std::vector<float> rgba(n * 4u, 1.0f);
        for (size_t i = 0; i < n; ++i) {
            rgba[i * 4 + 0] = linear_rgb[i * 3 + 0];
            rgba[i * 4 + 1] = linear_rgb[i * 3 + 1];
            rgba[i * 4 + 2] = linear_rgb[i * 3 + 2];
        }
        std::string error;
        if (!ptr::WriteExrMultilayer(path, rgba.data(), width, height, sample_counts, colorspace, &error)) {
            setErr(err, err_cap, error);
            return 1;
        }
        return 0;
    });
}

int ptr_host_write_exr_aovs(const char* path, const float* linear_rgb, const float* albedo_rgba, const float* normal_rgba, uint32_t width,
                            uint32_t height, char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!path || !linear_rgb || !albedo_rgba || !normal_rgba || width == 0 || height == 0) {
            setErr(err, err_cap, "ptr_host_write_exr_aovs: bad argument");
            return 1;
        }
        std::string error;
        if (!ptr::WriteExrAovs(path, linear_rgb, albedo_rgba, normal_rgba, width, height, &error)) {
            setErr(err, err_cap, error);
            return 1;
        }
        return 0;
    });
}

int ptr_host_decode_image(const uint8_t* data, uint64_t size, uint8_t* out_rgba, uint64_t cap_bytes, uint32_t* width, uint32_t* height, char* err,
                          size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!data || !width || !height) {
            setErr(err, err_cap, "ptr_host_decode_image: null argument");
            return 1;
        }
        ptr::DecodedImage img;
        std::string error;
        if (!ptr::DecodeImage(data, static_cast<size_t>(size), img, &error)) {
            setErr(err, err_cap, error);
            return 1;
        }
        *width = img.width;
        *height = img.height;
        if (out_rgba) {
            if (cap_bytes < img.rgba.size()) {
                setErr(err, err_cap, "ptr_host_decode_image: output buffer too small");
                return 1;
            }
            std::memcpy(out_rgba, img.rgba.data(), img.rgba.size());
        }
        return 0;
    });
}

int ptr_host_read_pfm(const char* path, float* out_rgb, uint32_t cap_floats, uint32_t* width, uint32_t* height) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return 1;
    char magic[3] = {0};
    unsigned w = 0, h = 0;
    float scale = 0.0f;
    if (std::fscanf(f, "%2s %u %u %f", magic, &w, &h, &scale) != 4 || std::strcmp(magic, "PF") != 0 || !(scale < 0.0f)) {
        std::fclose(f);
        return 1;
    }
    std::fgetc(f);  // single whitespace after the scale line
    if (width) *width = w;
    if (height) *height = h;
    const size_t need = static_cast<size_t>(w) * h * 3;
    if (!out_rgb || cap_floats < need) {
        std::fclose(f);
        return out_rgb ? 1 : 0;
    }
    bool ok = true;
    for (unsigned y = h; y-- > 0 && ok;) {
        ok = std::fread(out_rgb + static_cast<size_t>(y) * w * 3, sizeof(float), static_cast<size_t>(w) * 3, f) ==
             static_cast<size_t>(w) * 3;
    }
    std::fclose(f);
    return ok ? 0 : 1;
}

const char* ptr_version(void) { return "ptr-hip 0.1 (gfx950)"; }

int ptr_debug_env_distribution(const float* rgba, uint32_t w, uint32_t h, float* texel_pdf, uint32_t* cond_alias,
                               float* cond_threshold, uint32_t* marg_alias, float* marg_threshold, float* total_weight) {
    return guarded(nullptr, 0, [&]() -> int {
        ptr::EnvImportanceDistribution d;
        if (!ptr::BuildEnvImportanceDistribution(rgba, w, h, &d)) return 1;
        const size_t n = static_cast<size_t>(w) * h;
        for (size_t i = 0; i < n; ++i) {
            texel_pdf[i] = d.texelPdf[i];
            cond_alias[i] = d.conditional[i].alias;
            cond_threshold[i] = d.conditional[i].threshold;
        }
        for (uint32_t y = 0; y < h; ++y) {
            marg_alias[y] = d.marginal[y].alias;
            marg_threshold[y] = d.marginal[y].threshold;
        }
        if (total_weight) *total_weight = d.totalWeight;
        return 0;
    });
}

int ptr_debug_generate_tangents(const float* positions, const float* normals, const float* uvs, uint64_t triangle_count, float* out_tangents) {
    try {
        return ptr::GenerateTangentSpace(positions, normals, uvs, static_cast<size_t>(triangle_count), out_tangents) ? 0 : 1;
    } catch (...) {
        return 2;
    }
}

int ptr_debug_scene_geometry(const PtrSceneDesc* scene, uint32_t leaf_max, uint64_t out[16], char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!scene || !out) {
            if (err && err_cap) std::snprintf(err, err_cap, "ptr_debug_scene_geometry: null argument");
            return 1;
        }
        ptr::SceneGeometry geo;
        std::string error;
        if (!ptr::BuildSceneGeometry(*scene, leaf_max, geo, error)) {
            if (err && err_cap) std::snprintf(err, err_cap, "%s", error.c_str());
            return 1;
        }
        ptr::GeometryCheck c;
        ptr::ValidateSceneGeometry(geo, c);
        const float maxCell = std::max(std::max(geo.bvh.gridCell[0], geo.bvh.gridCell[1]), geo.bvh.gridCell[2]);
        const uint64_t vals[16] = {c.nodes, c.leaves, c.trianglesReferenced, c.spheresReferenced, c.maxDepth, c.maxLeafSize,
                                   c.unreferenced, c.multiplyReferenced, c.boxViolations, c.quantViolations, c.badRefs,
                                   geo.triCount, geo.sphereCount, static_cast<uint64_t>(geo.bvh.sahCost * 1000.0),
                                   static_cast<uint64_t>((geo.gatherSeconds + geo.buildSeconds + geo.flattenSeconds) * 1000.0),
                                   ((geo.bvh.nodeCount > 0 && maxCell * 8.0f <= geo.bvh.meanPrimExtent) ? 1u : 0u) | (c.oversize << 8) |
                                       (std::min<uint64_t>(c.wideNodes, 0xFFFFFFFFull) << 16) | (c.wideProblems ? 1ull << 63 : 0ull)};
        std::memcpy(out, vals, sizeof(vals));
        return 0;
    });
}

// Host-side walk of the scene's BVH with 1, 2 or 3 binary levels collapsed per step (two-, four-, eight-wide nodes: a child that is a
// leaf keeps its place), with four-wide nodes whose children are chosen by box area (levels = 4: chosen on the fly; levels = 5: the
// array BuildWideNodes ships, quantised boxes), children visited in order of entry distance: how many node
// steps, box tests and primitive tests a closest-hit query costs at each width.  Counts only - the product walks four-wide nodes on
// the device; this is the measurement behind DESIGN.md section 4.3c.
int ptr_debug_walk_counts(const PtrSceneDesc* scene, const float* rays, uint64_t n, uint32_t levels, uint64_t out[4], char* err, size_t err_cap) {
    return guarded(err, err_cap, [&]() -> int {
        if (!scene || (!rays && n) || !out || levels < 1u || levels > 5u) {
            setErr(err, err_cap, "ptr_debug_walk_counts: bad argument");
            return 1;
        }
        ptr::SceneGeometry geo;
        std::string error;
        if (!ptr::BuildSceneGeometry(*scene, 0, geo, error)) {
            setErr(err, err_cap, error);
            return 1;
        }
        const ptr::FlatBvh& bvh = geo.bvh;
        auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        struct Child { const float* lo; const float* hi; uint32_t ref; };
        uint64_t steps = 0, boxes = 0, prims = 0, hits = 0;
        std::vector<uint32_t> stack;
        // levels 5: the walk goes through the four-wide array BuildWideNodes makes (what the device walks), boxes as quantised
        std::unique_ptr<uint32_t[]> wideNodes;
        if (levels >= 5u) ptr::BuildWideNodes(bvh, ptr::WideCollapse::ByArea, wideNodes);
        for (uint64_t r = 0; r < n; ++r) {
            const float* q = rays + r * 8;
            const double o[3] = {q[0], q[1], q[2]}, d[3] = {q[4], q[5], q[6]};
            const double tmin = q[3];
            double tfar = q[7];
            bool hit = false;
            double inv[3];
            for (int a = 0; a < 3; ++a) inv[a] = 1.0 / (d[a] != 0.0 ? d[a] : 1e-300);
            auto testLeaf = [&](uint32_t ref) {
                const uint32_t first = ref & ptrk::kRefOffsetMask, count = ((ref >> ptrk::kRefCountShift) & 0xFu) % ptrk::kMaxLeafPrims + 1u;
                for (uint32_t k = 0; k < count; ++k) {
                    ++prims;
                    if (ref & ptrk::kRefSphereBit) {
                        const float* sp = &geo.sphereData[static_cast<size_t>(first + k) * 4];
                        double c[3], dd = 0, b = 0, cc = 0;
                        for (int a = 0; a < 3; ++a) { c[a] = sp[a] - o[a]; dd += d[a] * d[a]; b += c[a] * d[a]; cc += c[a] * c[a]; }
                        const double disc = b * b - dd * (cc - static_cast<double>(sp[3]) * sp[3]);
                        if (disc < 0) continue;
                        const double sq = std::sqrt(disc);
                        double t = (b - sq) / dd;
                        if (t < tmin) t = (b + sq) / dd;
                        if (t >= tmin && t <= tfar) { tfar = t; hit = true; }
                    } else {
                        const float* t3 = &geo.triData[static_cast<size_t>(first + k) * 12];
                        const double v0[3] = {t3[0], t3[1], t3[2]}, e1[3] = {t3[4], t3[5], t3[6]}, e2[3] = {t3[8], t3[9], t3[10]};   // e1 = v0 - v1, e2 = v2 - v0
                        const double ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
                        const double c[3] = {v0[0] - o[0], v0[1] - o[1], v0[2] - o[2]};
                        const double rr[3] = {c[1] * d[2] - c[2] * d[1], c[2] * d[0] - c[0] * d[2], c[0] * d[1] - c[1] * d[0]};
                        const double den = ng[0] * d[0] + ng[1] * d[1] + ng[2] * d[2];
                        if (den == 0.0) continue;
                        const double sgn = den < 0 ? -1.0 : 1.0, ad = std::fabs(den);
                        const double U = (rr[0] * e2[0] + rr[1] * e2[1] + rr[2] * e2[2]) * sgn, V = (rr[0] * e1[0] + rr[1] * e1[1] + rr[2] * e1[2]) * sgn;
                        if (U < 0 || V < 0 || U + V > ad) continue;
                        const double T = (ng[0] * c[0] + ng[1] * c[1] + ng[2] * c[2]) * sgn;
                        if (!(ad * tmin < T && T <= ad * tfar)) continue;
                        tfar = T / ad;
                        hit = true;
                    }
                }
            };
            stack.clear();
            if (bvh.oversizeRef != ptrk::kRefEmpty) testLeaf(bvh.oversizeRef);
            if (bvh.rootRef != ptrk::kRefEmpty) stack.push_back(bvh.rootRef);
            while (!stack.empty()) {
                const uint32_t ref = stack.back();
                stack.pop_back();
                if (ref & ptrk::kRefLeafBit) {
                    testLeaf(ref);
                    continue;
                }
                ++steps;
                if (levels >= 5u) {
                    const uint32_t* w = wideNodes.get() + static_cast<size_t>(ref) * 16u;
                    double entry[4];
                    uint32_t order[4], on = 0, refs[4];
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t* rec = w + 4u * k;
                        refs[k] = rec[3];
                        if (rec[3] == ptrk::kRefEmpty) continue;
                        ++boxes;
                        const double lo[3] = {bvh.gridOrigin[0] + double(rec[0] & 0xFFFFu) * bvh.gridCell[0], bvh.gridOrigin[1] + double(rec[0] >> 16) * bvh.gridCell[1],
                                              bvh.gridOrigin[2] + double(rec[1] & 0xFFFFu) * bvh.gridCell[2]};
                        const double hi[3] = {bvh.gridOrigin[0] + double(rec[1] >> 16) * bvh.gridCell[0], bvh.gridOrigin[1] + double(rec[2] & 0xFFFFu) * bvh.gridCell[1],
                                              bvh.gridOrigin[2] + double(rec[2] >> 16) * bvh.gridCell[2]};
                        double t0 = tmin, t1 = tfar;
                        for (int a = 0; a < 3; ++a) {
                            const double ta = (lo[a] - o[a]) * inv[a], tb = (hi[a] - o[a]) * inv[a];
                            t0 = std::max(t0, std::min(ta, tb));
                            t1 = std::min(t1, std::max(ta, tb));
                        }
                        if (t0 <= t1) {
                            entry[k] = t0;
                            order[on++] = k;
                        }
                    }
                    std::sort(order, order + on, [&](uint32_t a, uint32_t b) { return entry[a] > entry[b]; });
                    for (uint32_t k = 0; k < on; ++k) stack.push_back(refs[order[k]]);
                    continue;
                }
                Child kids[8];
                uint32_t kn = 0;
                auto children = [&](uint32_t node, Child* dst) {
                    const float* nd = &bvh.nodes[static_cast<size_t>(node) * 16];
                    uint32_t c = 0;
                    for (int sl = 0; sl < 2; ++sl) {
                        const uint32_t cref = bits(nd[sl == 0 ? 3 : 7]);
                        if (cref != ptrk::kRefEmpty) dst[c++] = Child{nd + sl * 8, nd + sl * 8 + 4, cref};
                    }
                    return c;
                };
                kn = children(ref, kids);
                if (levels == 4u) {
                    // four-wide by area: the internal child with the largest box is opened until four children stand (or only leaves do)
                    auto area = [](const Child& c) {
                        const double x = double(c.hi[0]) - c.lo[0], y = double(c.hi[1]) - c.lo[1], z = double(c.hi[2]) - c.lo[2];
                        return x * y + y * z + z * x;
                    };
                    while (kn < 4u) {
                        int best = -1;
                        for (uint32_t k = 0; k < kn; ++k) {
                            if (!(kids[k].ref & ptrk::kRefLeafBit) && (best < 0 || area(kids[k]) > area(kids[best]))) best = static_cast<int>(k);
                        }
                        if (best < 0) break;
                        Child two[2];
                        const uint32_t got = children(kids[best].ref, two);
                        kids[best] = two[0];
                        if (got > 1u) kids[kn++] = two[1];
                    }
                }
                for (uint32_t l = 1; l < levels && levels != 4u; ++l) {   // replace every internal child by its own children
                    Child next[8];
                    uint32_t nn = 0;
                    for (uint32_t k = 0; k < kn; ++k) {
                        if (kids[k].ref & ptrk::kRefLeafBit) next[nn++] = kids[k];
                        else nn += children(kids[k].ref, next + nn);
                    }
                    std::memcpy(kids, next, sizeof(Child) * nn);
                    kn = nn;
                }
                double entry[8];
                uint32_t order[8], on = 0;
                for (uint32_t k = 0; k < kn; ++k) {
                    ++boxes;
                    double t0 = tmin, t1 = tfar;
                    for (int a = 0; a < 3; ++a) {
                        const double ta = (kids[k].lo[a] - o[a]) * inv[a], tb = (kids[k].hi[a] - o[a]) * inv[a];
                        t0 = std::max(t0, std::min(ta, tb));
                        t1 = std::min(t1, std::max(ta, tb));
                    }
                    if (t0 <= t1) {
                        entry[k] = t0;
                        order[on++] = k;
                    }
                }
                std::sort(order, order + on, [&](uint32_t a, uint32_t b) { return entry[a] > entry[b]; });   // far first: the nearest is popped next
                for (uint32_t k = 0; k < on; ++k) stack.push_back(kids[order[k]].ref);
            }
            hits += hit ? 1u : 0u;
        }
        out[0] = steps;
        out[1] = boxes;
        out[2] = prims;
        out[3] = hits;
        return 0;
    });
}

}  // extern "C"
