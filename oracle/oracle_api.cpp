// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md.
// C entry points of liboracle.so, used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include <chrono>
#include <cstring>
#include <memory>

#include "oracle_integrator.h"

using namespace oracle;

struct OracleScene {
    Scene scene;
    PtrSceneDesc desc;  // borrowed pointers: caller keeps the arrays alive
};

extern "C" {

OracleScene* oracle_scene_create(const PtrSceneDesc* desc) {
    if (!desc) return nullptr;
    auto s = std::make_unique<OracleScene>();
    s->desc = *desc;
    s->scene.build(*desc);
    return s.release();
}

void oracle_scene_destroy(OracleScene* s) { delete s; }

// info: [0]=prims [1]=bvh nodes [2]=geoms
void oracle_scene_info(const OracleScene* s, uint64_t out[4]) {
    out[0] = s->scene.prims.size();
    out[1] = s->scene.nodes.size();
    out[2] = s->scene.geoms.size();
    out[3] = 0;
}

// Renders rows [y0,y1) into out_rgb (full W*H*3 buffer).  counters (optional): extendRays, shadowRays,
// nodes, prims, shadedHits, triangleHits.  Returns seconds spent in the tile loop.
double oracle_render(const OracleScene* s, const PtrSettings* settings, uint32_t spp, uint32_t threads,
                     uint32_t y0, uint32_t y1, float* out_rgb, uint64_t* counters) {
    RenderCounters rc;
    const auto t0 = std::chrono::steady_clock::now();
    render(s->scene, s->desc, *settings, spp, threads, y0, y1, out_rgb, counters ? &rc : nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    if (counters) {
        counters[0] = rc.extendRays;
        counters[1] = rc.shadowRays;
        counters[2] = rc.nodes;
        counters[3] = rc.prims;
        counters[4] = rc.shadedHits;
        counters[5] = rc.triangleHits;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

// As oracle_render, plus per-pixel path signatures and "marginal shadow decision" flags (W*H each; see render()).
double oracle_render_signatures(const OracleScene* s, const PtrSettings* settings, uint32_t spp, uint32_t threads, uint32_t y0,
                                uint32_t y1, float* out_rgb, uint32_t* out_signature, uint8_t* out_marginal) {
    const auto t0 = std::chrono::steady_clock::now();
    render(s->scene, s->desc, *settings, spp, threads, y0, y1, out_rgb, nullptr, out_signature, out_marginal);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Texture filtering rule (oracle_integrator.cpp "material textures"): in: n * 3 floats {u, v, lod}; out: n * 4 floats RGBA
void oracle_texture_sample(const PtrSceneDesc* desc, uint32_t texture, const float* in, uint64_t n, float* out) {
    sampleTextures(*desc, texture, in, n, out);
}

// rays: n * 8 floats {ox,oy,oz,tmin,dx,dy,dz,tmax}
void oracle_trace_rays(const OracleScene* s, const float* rays, uint64_t n, int any_hit, int brute_force, PtrHit* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = rays + i * 8;
        const V3 org(r[0], r[1], r[2]), dir(r[4], r[5], r[6]);
        PtrHit h;
        std::memset(&h, 0, sizeof(h));
        h.t = -1.0f;
        if (any_hit) {
            if (s->scene.occluded(org, dir, r[3], r[7], brute_force != 0)) h.t = 0.0f;
        } else {
            RayHit rh;
            if (s->scene.intersect(org, dir, r[3], r[7], rh, brute_force != 0)) {
                const Geom& g = s->scene.geoms[rh.geom];
                h.t = rh.t;
                h.u = rh.u;
                h.v = rh.v;
                h.ng[0] = rh.ng.x;
                h.ng[1] = rh.ng.y;
                h.ng[2] = rh.ng.z;
                if (g.type == GeomType::Mesh) {
                    h.primType = 0;
                    h.geomIndex = g.meshIndex;
                    h.primIndex = rh.primId;
                } else if (g.type == GeomType::Spheres) {
                    h.primType = 1;
                    h.primIndex = rh.primId;
                } else {
                    h.primType = 2;
                    h.primIndex = rh.primId < g.triToRect.size() ? g.triToRect[rh.primId] : rh.primId;
                }
            }
        }
        out[i] = h;
    }
}

// Surface records of the integrator's IntersectScene twin (E:2302-2410: position, geometric and interpolated shading normal, facing)
// and the next-ray origin OffsetRayOrigin gives for a direction (E:917-931).
// in: n * 9 floats {origin, direction, next direction}; out: n * 16 floats {hit, t, position, normal, shading normal, front face, next origin}
void oracle_surface_hits(const OracleScene* s, const float* in, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = in + i * 9;
        float* o = out + i * 16;
        std::memset(o, 0, 16 * sizeof(float));
        HitInfo hit;
        const Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5])};
        if (!intersectScene(s->scene, ray, hit)) continue;
        const V3 next = nextRayOrigin(hit, V3(r[6], r[7], r[8]));
        o[0] = 1.0f;
        o[1] = hit.t;
        o[2] = hit.position.x, o[3] = hit.position.y, o[4] = hit.position.z;
        o[5] = hit.normal.x, o[6] = hit.normal.y, o[7] = hit.normal.z;
        o[8] = hit.shadingNormal.x, o[9] = hit.shadingNormal.y, o[10] = hit.shadingNormal.z;
        o[11] = hit.frontFace ? 1.0f : 0.0f;
        o[12] = next.x, o[13] = next.y, o[14] = next.z;
    }
}

// ---- known-answer helpers ----

uint32_t oracle_rng_hash(uint32_t x) { return Rng::hash(x); }

void oracle_rng_floats(uint32_t state, uint32_t n, float* out, uint32_t* out_state) {
    Rng rng;
    rng.state = state;
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.nextFloat();
    if (out_state) *out_state = rng.state;
}

// out: origin, lowerLeft, horizontal, vertical, u, v (18 floats) + lensRadius
void oracle_build_camera(const PtrSettings* s, float out[19]) {
    const Camera c = buildCamera(*s);
    const V3 v[6] = {c.origin, c.lowerLeft, c.horizontal, c.vertical, c.u, c.v};
    for (int i = 0; i < 6; ++i) {
        out[i * 3 + 0] = v[i].x;
        out[i * 3 + 1] = v[i].y;
        out[i * 3 + 2] = v[i].z;
    }
    out[18] = c.lensRadius;
}

// Primary rays for (x,y,sample) triples exactly as the render loop draws them. out: n*6 floats (origin, dir), states: rng state after
void oracle_camera_rays(const PtrSettings* s, const uint32_t* xys, uint64_t n, float* out, uint32_t* out_states) {
    const Camera c = buildCamera(*s);
    const uint32_t seedBase = s->seed != 0 ? s->seed : 0x9e3779b9u;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t x = xys[i * 3], y = xys[i * 3 + 1], smp = xys[i * 3 + 2];
        Rng rng;
        rng.state = Rng::hash(seedBase ^ (y * s->width + x) ^ (smp * 0x9e3779b9u));
        const Ray r = generateCameraRay(c, s->width, s->height, x, y, rng);
        float* o = out + i * 6;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        if (out_states) out_states[i] = rng.state;
    }
}

// in: n * 12 floats {position, normal, wo, wi}; out: n * 5 floats {value rgb, pdf, isDelta}
void oracle_eval_bsdf(const PtrMaterial* m, const PtrSettings* s, const float* in, uint64_t n, float* out) {
    const ClampParams cp = makeClampParams(*s);
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = in + i * 12;
        const BsdfEval e = evaluateBsdf(*m, V3(p), V3(p + 3), V3(p + 6), V3(p + 9), cp);
        float* o = out + i * 5;
        o[0] = e.value.x; o[1] = e.value.y; o[2] = e.value.z; o[3] = e.pdf; o[4] = e.isDelta ? 1.0f : 0.0f;
    }
}

// in: n * 9 floats {position, normal, wo} (incident = -wo), front_face/rng_state per item;
// out: n * 8 floats {direction, weight, pdf, isDelta}, out_states: rng state after sampling
void oracle_sample_bsdf(const PtrMaterial* m, const PtrSettings* s, const float* in, const uint32_t* front_face,
                        const uint32_t* rng_states, uint64_t n, float* out, uint32_t* out_states) {
    const ClampParams cp = makeClampParams(*s);
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = in + i * 9;
        Rng rng;
        rng.state = rng_states[i];
        const V3 wo(p + 6);
        const BsdfSample b = sampleBsdf(*m, V3(p), V3(p + 3), wo, -wo, front_face[i] != 0, rng, cp);
        float* o = out + i * 8;
        o[0] = b.direction.x; o[1] = b.direction.y; o[2] = b.direction.z;
        o[3] = b.weight.x; o[4] = b.weight.y; o[5] = b.weight.z;
        o[6] = b.pdf; o[7] = b.isDelta ? 1.0f : 0.0f;
        if (out_states) out_states[i] = rng.state;
    }
}

// Alias-table build: outputs sized by caller (texelPdf/condAlias/condThreshold: w*h; margAlias/margThreshold: h).
int oracle_env_build(const float* rgba, uint32_t w, uint32_t h, float* texel_pdf, uint32_t* cond_alias,
                     float* cond_threshold, uint32_t* marg_alias, float* marg_threshold, float* total_weight) {
    EnvDistribution d;
    if (!buildEnvDistribution(rgba, w, h, d)) return 1;
    const size_t n = static_cast<size_t>(w) * h;
    std::memcpy(texel_pdf, d.texelPdf.data(), n * sizeof(float));
    std::memcpy(cond_alias, d.conditionalAlias.data(), n * sizeof(uint32_t));
    std::memcpy(cond_threshold, d.conditionalThreshold.data(), n * sizeof(float));
    std::memcpy(marg_alias, d.marginalAlias.data(), h * sizeof(uint32_t));
    std::memcpy(marg_threshold, d.marginalThreshold.data(), h * sizeof(float));
    if (total_weight) *total_weight = d.totalWeight;
    return 0;
}

// u: n*3 randoms; out: n*7 {direction, radiance, pdf}; lookups: n*4 {SampleEnvironment(dir) rgb, EnvironmentPdf(dir)}
int oracle_env_sample(const float* rgba, uint32_t w, uint32_t h, float rotation, float intensity, const float* u,
                      uint64_t n, float* out, float* lookups) {
    EnvMap env;
    env.rgba = rgba;
    env.width = w;
    env.height = h;
    env.hasDistribution = buildEnvDistribution(rgba, w, h, env.dist);
    if (!env.hasDistribution) return 1;
    for (uint64_t i = 0; i < n; ++i) {
        const EnvSample s = sampleEnvironmentCpu(env.dist, u[i * 3], u[i * 3 + 1], u[i * 3 + 2], rotation, intensity, rgba);
        float* o = out + i * 7;
        o[0] = s.direction.x; o[1] = s.direction.y; o[2] = s.direction.z;
        o[3] = s.radiance.x; o[4] = s.radiance.y; o[5] = s.radiance.z; o[6] = s.pdf;
        if (lookups) {
            const V3 c = sampleEnvironment(env, s.direction, rotation, intensity);
            float* l = lookups + i * 4;
            l[0] = c.x; l[1] = c.y; l[2] = c.z;
            l[3] = environmentPdf(env, rotation, s.direction);
        }
    }
    return 0;
}

}  // extern "C"
