// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md.
//
// Driver around the reference's own BVH dependency, tinybvh 1.6.7 (header-only, /root/reference/external/tinybvh/tiny_bvh.h,
// version at :92-94), compiled from where it lies by oracle/Makefile into oracle/_ref/libref_tinybvh.so.  It does what the
// reference's software path does with the library - `tinybvh::BVH::Build(bvhvec4*, triCount)` on three vertices per triangle
// (src/renderer/SceneAccel.mm:104-106) - and exposes the library's OWN closest-hit / occlusion traversal, so the tests can
// hold the oracle's scalar ray caster (and the HIP path) against an intersector that is the reference's code, not a
// restatement: hit distances within float tolerance, the same primitive except on ties, and a tree of comparable quality to
// this build's binned-SAH builder.  Nothing here is written by hand except this glue; no reference source is copied.
#include <cstdint>
#include <cstring>
#include <vector>

#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"

struct RefBvh {
    std::vector<tinybvh::bvhvec4> verts;
    tinybvh::BVH bvh;
};

extern "C" {

// positions: tri_count * 9 floats (v0, v1, v2 per triangle, world space)
RefBvh* ref_bvh_build(const float* positions, uint32_t tri_count) {
    RefBvh* r = new RefBvh();
    r->verts.resize(static_cast<size_t>(tri_count) * 3u);
    for (size_t i = 0; i < r->verts.size(); ++i) r->verts[i] = tinybvh::bvhvec4(positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], 0.0f);
    r->bvh.Build(r->verts.data(), tri_count);
    return r;
}

void ref_bvh_free(RefBvh* r) { delete r; }

// out[0] = nodes in use (tinybvh leaves index 1 unused), out[1] = leaves, out[2] = SAH cost * 1000, out[3] = library version
void ref_bvh_info(const RefBvh* r, uint64_t out[4]) {
    uint64_t leaves = 0;
    for (uint32_t n = 0; n < r->bvh.usedNodes; ++n) {
        if (n == 1) continue;
        if (r->bvh.bvhNode[n].triCount > 0) ++leaves;
    }
    out[0] = static_cast<uint64_t>(r->bvh.NodeCount());
    out[1] = leaves;
    out[2] = static_cast<uint64_t>(r->bvh.SAHCost() * 1000.0f);
    out[3] = TINY_BVH_VERSION_MAJOR * 10000u + TINY_BVH_VERSION_MINOR * 100u + TINY_BVH_VERSION_SUB;
}

// rays: n * 8 floats {ox,oy,oz,tmin(ignored: the library starts at 0),dx,dy,dz,tmax}; directions must be unit length.
// out_t: hit distance or -1; out_prim: triangle index or 0xFFFFFFFF; out_uv: n * 2 barycentrics
void ref_bvh_intersect(const RefBvh* r, const float* rays, uint64_t n, float* out_t, uint32_t* out_prim, float* out_uv) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + i * 8;
        tinybvh::Ray ray(tinybvh::bvhvec3(q[0], q[1], q[2]), tinybvh::bvhvec3(q[4], q[5], q[6]), q[7]);
        r->bvh.Intersect(ray);
        const bool hit = ray.hit.t < q[7];
        out_t[i] = hit ? ray.hit.t : -1.0f;
        out_prim[i] = hit ? ray.hit.prim : 0xFFFFFFFFu;
        if (out_uv) {
            out_uv[i * 2] = hit ? ray.hit.u : 0.0f;
            out_uv[i * 2 + 1] = hit ? ray.hit.v : 0.0f;
        }
    }
}

void ref_bvh_occluded(const RefBvh* r, const float* rays, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + i * 8;
        const tinybvh::Ray ray(tinybvh::bvhvec3(q[0], q[1], q[2]), tinybvh::bvhvec3(q[4], q[5], q[6]), q[7]);
        out[i] = r->bvh.IsOccluded(ray) ? 1u : 0u;
    }
}

}  // extern "C"
