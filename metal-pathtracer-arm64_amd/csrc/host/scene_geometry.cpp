#include "scene_geometry.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "../kernels/bvh_layout.h"
#include "knobs.h"
#include "parallel.h"
#include "vecmath.h"

namespace ptr {
namespace {

struct M4 {
    float m[4][4];  // m[col][row]
};

M4 loadM4(const float* p) {
    M4 r;
    std::memcpy(r.m, p, sizeof(r.m));
    return r;
}

// Cofactor inverse in float (the Embree backend calls simd_inverse for the normal matrix).
M4 inverse(const M4& a) {
    const float* s = &a.m[0][0];
    float c[16];
    auto d3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        auto e = [&](int r, int col) { return s[col * 4 + r]; };
        return e(r0, c0) * (e(r1, c1) * e(r2, c2) - e(r1, c2) * e(r2, c1)) - e(r0, c1) * (e(r1, c0) * e(r2, c2) - e(r1, c2) * e(r2, c0)) +
               e(r0, c2) * (e(r1, c0) * e(r2, c1) - e(r1, c1) * e(r2, c0));
    };
    const int idx[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    for (int row = 0; row < 4; ++row) {
        for (int col = 0; col < 4; ++col) {
            const float minor = d3(idx[row][0], idx[row][1], idx[row][2], idx[col][0], idx[col][1], idx[col][2]);
            c[row * 4 + col] = ((row + col) & 1) ? -minor : minor;   // cofactor of element (row, col)
        }
    }
    float det = 0.0f;
    for (int col = 0; col < 4; ++col) det += s[col * 4 + 0] * c[0 * 4 + col];
    M4 r;
    const float invDet = 1.0f / det;
    // inverse(row, col) = cofactor(col, row) / det; stored column-major
    for (int col = 0; col < 4; ++col) {
        for (int row = 0; row < 4; ++row) r.m[col][row] = c[col * 4 + row] * invDet;
    }
    return r;
}

float3 transformPoint(const M4& t, const float* p) {
    return {((t.m[0][0] * p[0] + t.m[1][0] * p[1]) + t.m[2][0] * p[2]) + t.m[3][0],
            ((t.m[0][1] * p[0] + t.m[1][1] * p[1]) + t.m[2][1] * p[2]) + t.m[3][1],
            ((t.m[0][2] * p[0] + t.m[1][2] * p[1]) + t.m[2][2] * p[2]) + t.m[3][2]};
}

void padBounds(BuildPrim& p) {
    for (int a = 0; a < 3; ++a) {
        const float pad = 1e-5f * std::max(std::max(std::fabs(p.lo[a]), std::fabs(p.hi[a])), 1.0f);
        p.lo[a] -= pad;
        p.hi[a] += pad;
    }
}

float bitsToFloat(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

uint32_t floatBits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// Splits [0, n) into contiguous chunks over the host threads (large meshes only; small ones stay serial).
void parallelFor(size_t n, const std::function<void(size_t, size_t)>& body) {
    const size_t kSerialBelow = 1u << 16;
    unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n < kSerialBelow || threads == 1) {
        body(0, n);
        return;
    }
    const size_t chunk = (n + threads - 1) / threads;
    runOnThreads(threads, [&](uint32_t t) {
        const size_t b = std::min(n, chunk * t), e = std::min(n, chunk * (t + 1));
        if (b < e) body(b, e);
    });
}

// Writes one triangle (input order) into the 12-float records and its padded bounds.
void storeTri(float* tri, float* nrm, BuildPrim& prim, const float3& v0, const float3& v1, const float3& v2, const float3& n0,
              const float3& n1, const float3& n2, uint32_t material, uint32_t shadeKey, uint32_t kind, uint32_t geomIndex, uint32_t primIndex) {
    const float3 e1 = v0 - v1, e2 = v2 - v0;
    tri[0] = v0.x, tri[1] = v0.y, tri[2] = v0.z, tri[3] = bitsToFloat(material);
    tri[4] = e1.x, tri[5] = e1.y, tri[6] = e1.z, tri[7] = bitsToFloat((kind << 30) | ((shadeKey & ptrk::kHitKeyMask) << ptrk::kHitKeyShift) | (geomIndex & ptrk::kTriGeomMask));
    tri[8] = e2.x, tri[9] = e2.y, tri[10] = e2.z, tri[11] = bitsToFloat(primIndex);
    nrm[0] = n0.x, nrm[1] = n0.y, nrm[2] = n0.z, nrm[3] = 0.0f;
    nrm[4] = n1.x, nrm[5] = n1.y, nrm[6] = n1.z, nrm[7] = 0.0f;
    nrm[8] = n2.x, nrm[9] = n2.y, nrm[10] = n2.z, nrm[11] = 0.0f;
    const float3* vs[3] = {&v0, &v1, &v2};
    for (int a = 0; a < 3; ++a) {
        prim.lo[a] = std::min(std::min((&vs[0]->x)[a], (&vs[1]->x)[a]), (&vs[2]->x)[a]);
        prim.hi[a] = std::max(std::max((&vs[0]->x)[a], (&vs[1]->x)[a]), (&vs[2]->x)[a]);
    }
    prim.isSphere = 0;
    padBounds(prim);
}

double secondsSince(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

bool BuildSceneGeometry(const PtrSceneDesc& desc, uint32_t leafMax, SceneGeometry& out, std::string& error) {
    out = SceneGeometry{};
    auto t0 = std::chrono::steady_clock::now();

    uint64_t triTotal = static_cast<uint64_t>(desc.rectCount) * 2u;
    for (uint32_t mi = 0; mi < desc.meshCount; ++mi) {
        const PtrMeshDesc& mesh = desc.meshes[mi];
        if (mesh.vertexCount == 0 || mesh.indexCount == 0) continue;
        triTotal += mesh.indexCount / 3u;
    }
    if (triTotal + desc.sphereCount > ptrk::kRefOffsetMask) {
        error = "scene exceeds 64M primitives";
        return false;
    }
    if (desc.meshCount > ptrk::kTriGeomMask) {
        error = "scene exceeds 64M meshes";
        return false;
    }
    // shade key of a primitive (kernels/bvh_layout.h): material type + 1, what k_shade sorts the slots of a block by
    auto shadeKeyOf = [&](uint32_t material) -> uint32_t {
        if (desc.materialCount == 0u || desc.materials == nullptr) return 0u;
        const uint32_t type = static_cast<uint32_t>(desc.materials[std::min(material, desc.materialCount - 1u)].typeEta[0]);
        return std::min(type, 7u) + 1u;
    };
    const uint32_t triCount = static_cast<uint32_t>(triTotal);
    std::vector<float> triIn(static_cast<size_t>(triCount) * 12), nrmIn(static_cast<size_t>(triCount) * 12);
    std::vector<BuildPrim> prims(static_cast<size_t>(triCount) + desc.sphereCount);
    // texture attributes, input order (textured scenes only): see SceneGeometry::triUv / triTangent
    bool textured = false;
    if (desc.textures && desc.textureCount > 0) {
        for (uint32_t mi = 0; mi < desc.meshCount; ++mi) textured = textured || desc.meshes[mi].uv0 != nullptr || desc.meshes[mi].uv1 != nullptr;
    }
    std::vector<float> uvIn, tanIn;
    if (textured) {
        uvIn.assign(static_cast<size_t>(triCount) * 16, 0.0f);
        tanIn.assign(static_cast<size_t>(triCount) * 12, 0.0f);
    }

    // meshes: baked to world space, normals through the inverse-transpose (EmbreeHeadlessRenderer.mm:2100-2166)
    size_t cursor = 0;
    std::atomic<bool> badIndex{false};
    for (uint32_t mi = 0; mi < desc.meshCount; ++mi) {
        const PtrMeshDesc& mesh = desc.meshes[mi];
        if (mesh.vertexCount == 0 || mesh.indexCount == 0) continue;
        const M4 l2w = loadM4(mesh.localToWorld);
        const M4 w2l = inverse(l2w);
        const float det3 = l2w.m[0][0] * (l2w.m[1][1] * l2w.m[2][2] - l2w.m[2][1] * l2w.m[1][2]) -
                           l2w.m[1][0] * (l2w.m[0][1] * l2w.m[2][2] - l2w.m[2][1] * l2w.m[0][2]) +
                           l2w.m[2][0] * (l2w.m[0][1] * l2w.m[1][2] - l2w.m[1][1] * l2w.m[0][2]);
        const float detSign = det3 < 0.0f ? -1.0f : 1.0f;
        const float3 nc0{w2l.m[0][0], w2l.m[1][0], w2l.m[2][0]};
        const float3 nc1{w2l.m[0][1], w2l.m[1][1], w2l.m[2][1]};
        const float3 nc2{w2l.m[0][2], w2l.m[1][2], w2l.m[2][2]};
        std::vector<float3> pos(mesh.vertexCount), nrm(mesh.vertexCount);
        parallelFor(mesh.vertexCount, [&](size_t b, size_t e) {
            for (size_t v = b; v < e; ++v) {
                pos[v] = transformPoint(l2w, mesh.positions + 3 * v);
                const float* n = mesh.normals + 3 * v;
                const float3 wn = (nc0 * n[0] + nc1 * n[1]) + nc2 * n[2];
                nrm[v] = length(wn) > 0.0f ? normalize(wn) : wn;
            }
        });
        const size_t meshTris = mesh.indexCount / 3u;
        const size_t base = cursor;
        parallelFor(meshTris, [&](size_t b, size_t e) {
            for (size_t t = b; t < e; ++t) {
                const uint32_t i0 = mesh.indices[3 * t], i1 = mesh.indices[3 * t + 1], i2 = mesh.indices[3 * t + 2];
                if (i0 >= mesh.vertexCount || i1 >= mesh.vertexCount || i2 >= mesh.vertexCount) {
                    badIndex.store(true);
                    return;
                }
                const size_t k = base + t;
                storeTri(&triIn[k * 12], &nrmIn[k * 12], prims[k], pos[i0], pos[i1], pos[i2], nrm[i0], nrm[i1], nrm[i2],
                         mesh.materialIndex, shadeKeyOf(mesh.materialIndex), 0u, mi, static_cast<uint32_t>(t));
                if (textured) {
                    const uint32_t vi[3] = {i0, i1, i2};
                    float* uv = &uvIn[k * 16];
                    float* tg = &tanIn[k * 12];
                    for (int c = 0; c < 3; ++c) {
                        if (mesh.uv0) {
                            uv[c * 4 + 0] = mesh.uv0[2 * vi[c]];
                            uv[c * 4 + 1] = mesh.uv0[2 * vi[c] + 1];
                        }
                        if (mesh.uv1) {
                            uv[c * 4 + 2] = mesh.uv1[2 * vi[c]];
                            uv[c * 4 + 3] = mesh.uv1[2 * vi[c] + 1];
                        }
                        if (mesh.tangents) {
                            // world-space tangent: the linear part of localToWorld (interpolate_tangent, shaders/pathtrace.metal:693-739)
                            const float* tl = mesh.tangents + 4 * vi[c];
                            tg[c * 4 + 0] = (l2w.m[0][0] * tl[0] + l2w.m[1][0] * tl[1]) + l2w.m[2][0] * tl[2];
                            tg[c * 4 + 1] = (l2w.m[0][1] * tl[0] + l2w.m[1][1] * tl[1]) + l2w.m[2][1] * tl[2];
                            tg[c * 4 + 2] = (l2w.m[0][2] * tl[0] + l2w.m[1][2] * tl[1]) + l2w.m[2][2] * tl[2];
                            tg[c * 4 + 3] = tl[3] == 0.0f ? 0.0f : (tl[3] < 0.0f ? -1.0f : 1.0f) * detSign;
                        }
                    }
                    // uv-per-world of both sets (triangle_surface_partials, shaders/pathtrace.metal:741-820) in world space
                    const float3 edge1 = pos[i1] - pos[i0], edge2 = pos[i2] - pos[i0];
                    for (int set = 0; set < 2; ++set) {
                        const float du1 = uv[4 + set * 2] - uv[set * 2], dv1 = uv[4 + set * 2 + 1] - uv[set * 2 + 1];
                        const float du2 = uv[8 + set * 2] - uv[set * 2], dv2 = uv[8 + set * 2 + 1] - uv[set * 2 + 1];
                        const float det = du1 * dv2 - dv1 * du2;
                        float perWorld = 0.0f;
                        bool done = false;
                        if (std::fabs(det) > 1.0e-9f) {
                            const float inv = 1.0f / det;
                            const float3 dPdu = (edge1 * dv2 - edge2 * dv1) * inv, dPdv = (edge2 * du1 - edge1 * du2) * inv;
                            const float lenU = length(dPdu), lenV = length(dPdv);
                            if (lenU > 1.0e-8f && lenV > 1.0e-8f) {
                                perWorld = std::max(1.0f / lenU, 1.0f / lenV);
                                done = std::isfinite(perWorld) && perWorld > 0.0f;
                            }
                        }
                        if (!done) {
                            const float worldArea = length(cross(edge1, edge2)), uvArea = std::fabs(det);
                            perWorld = (worldArea > 1.0e-12f && uvArea > 1.0e-12f) ? std::sqrt(uvArea / worldArea) : 0.0f;
                            if (!std::isfinite(perWorld)) perWorld = 0.0f;
                        }
                        uv[12 + set] = perWorld;
                    }
                    uv[14] = detSign;
                }
            }
        });
        cursor += meshTris;
    }
    if (badIndex.load()) {
        error = "mesh index out of range";
        return false;
    }
    // rectangles: two triangles each, winding chosen to agree with the stored normal (:2211-2257)
    for (uint32_t ri = 0; ri < desc.rectCount; ++ri) {
        const PtrRect& r = desc.rects[ri];
        const float3 c{r.corner[0], r.corner[1], r.corner[2]}, eu{r.edgeU[0], r.edgeU[1], r.edgeU[2]},
            ev{r.edgeV[0], r.edgeV[1], r.edgeV[2]};
        const float3 n = normalize(float3{r.normalAndPlane[0], r.normalAndPlane[1], r.normalAndPlane[2]});
        const float3 p[4] = {c, c + eu, c + ev, (c + eu) + ev};
        const bool flip = dot(normalize(cross(eu, ev)), n) < 0.0f;
        const int order[2][6] = {{0, 1, 2, 2, 1, 3}, {0, 2, 1, 1, 2, 3}};
        const int* o = order[flip ? 1 : 0];
        for (int half = 0; half < 2; ++half) {
            const size_t k = cursor++;
            storeTri(&triIn[k * 12], &nrmIn[k * 12], prims[k], p[o[half * 3]], p[o[half * 3 + 1]], p[o[half * 3 + 2]], n, n, n,
                     r.materialTwoSided[0], shadeKeyOf(r.materialTwoSided[0]), 2u, ri, ri * 2u + static_cast<uint32_t>(half));
        }
    }
    for (uint32_t si = 0; si < desc.sphereCount; ++si) {
        const PtrSphere& s = desc.spheres[si];
        BuildPrim& p = prims[static_cast<size_t>(triCount) + si];
        const float rad = std::fabs(s.centerRadius[3]);
        for (int a = 0; a < 3; ++a) {
            p.lo[a] = s.centerRadius[a] - rad;
            p.hi[a] = s.centerRadius[a] + rad;
        }
        p.isSphere = 1;
        padBounds(p);
    }
    out.gatherSeconds = secondsSince(t0);

    t0 = std::chrono::steady_clock::now();
    if (leafMax == 0) {
        leafMax = 4;
    }
    BuildFlatBvh(prims, out.bvh, 0, leafMax);
    out.buildSeconds = secondsSince(t0);

    // leaf-order arrays
    t0 = std::chrono::steady_clock::now();
    out.triCount = triCount;
    out.sphereCount = desc.sphereCount;
    out.triData.resize(static_cast<size_t>(triCount) * 12);
    out.triNormals.resize(static_cast<size_t>(triCount) * 12);
    const std::vector<uint32_t>& order = out.bvh.triOrder;
    parallelFor(order.size(), [&](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k) {
            std::memcpy(&out.triData[k * 12], &triIn[static_cast<size_t>(order[k]) * 12], 48);
            std::memcpy(&out.triNormals[k * 12], &nrmIn[static_cast<size_t>(order[k]) * 12], 48);
        }
    });
    // where the two halves of every rectangle ended up (k_shade tests a sampled light's own triangles before queueing a shadow ray)
    out.rectTriLeaf.assign(static_cast<size_t>(desc.rectCount) * 2u, 0xFFFFFFFFu);
    {
        const size_t meshTriTotal = static_cast<size_t>(triCount) - static_cast<size_t>(desc.rectCount) * 2u;   // rectangle halves follow the meshes in input order
        for (size_t k = 0; k < order.size(); ++k) {
            if (order[k] >= meshTriTotal) out.rectTriLeaf[order[k] - meshTriTotal] = static_cast<uint32_t>(k);
        }
    }
    if (textured) {
        out.triUv.resize(static_cast<size_t>(triCount) * 16);
        out.triTangent.resize(static_cast<size_t>(triCount) * 12);
        parallelFor(order.size(), [&](size_t b, size_t e) {
            for (size_t k = b; k < e; ++k) {
                std::memcpy(&out.triUv[k * 16], &uvIn[static_cast<size_t>(order[k]) * 16], 64);
                std::memcpy(&out.triTangent[k * 12], &tanIn[static_cast<size_t>(order[k]) * 12], 48);
            }
        });
    }
    for (uint32_t idx : out.bvh.sphereOrder) {
        const PtrSphere& s = desc.spheres[idx];
        out.sphereData.insert(out.sphereData.end(), s.centerRadius, s.centerRadius + 4);
        out.sphereInfo.push_back(idx);
        out.sphereInfo.push_back(s.materialIndex[0]);
    }
    out.flattenSeconds = secondsSince(t0);
    if (readKnobs().verboseBuild) {
        std::fprintf(stderr, "[geometry] gather %.2f s, bvh %.2f s, leaf-order %.2f s (%u triangles, %u spheres)\n", out.gatherSeconds,
                     out.buildSeconds, out.flattenSeconds, out.triCount, out.sphereCount);
    }
    return true;
}

void ValidateSceneGeometry(const SceneGeometry& g, GeometryCheck& out) {
    out = GeometryCheck{};
    const FlatBvh& bvh = g.bvh;
    out.nodes = bvh.nodeCount;
    const size_t nodeCount = bvh.nodeCount;
    std::vector<uint8_t> triSeen(g.triCount, 0), sphSeen(g.sphereCount, 0);
    struct Box {
        float lo[3], hi[3];
    };
    const float inf = INFINITY;
    std::vector<Box> subtree(nodeCount, Box{{inf, inf, inf}, {-inf, -inf, -inf}});
    std::vector<uint32_t> depth(nodeCount, 0);
    if (nodeCount > 0) depth[0] = 1;

    auto grow = [](Box& b, const float* p) {
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::min(b.lo[a], p[a]);
            b.hi[a] = std::max(b.hi[a], p[a]);
        }
    };
    auto leafBounds = [&](uint32_t ref, Box& b) {
        const uint32_t first = ref & ptrk::kRefOffsetMask;
        const uint32_t count = ((ref >> ptrk::kRefCountShift) & 0xFu) + 1u;
        out.leaves += 1;
        out.maxLeafSize = std::max<uint64_t>(out.maxLeafSize, count);
        const bool sphere = (ref & ptrk::kRefSphereBit) != 0u;
        for (uint32_t i = first; i < first + count; ++i) {
            if (sphere) {
                if (i >= g.sphereCount) {
                    out.badRefs += 1;
                    continue;
                }
                if (sphSeen[i]++) out.multiplyReferenced += 1;
                out.spheresReferenced += 1;
                const float* s = &g.sphereData[static_cast<size_t>(i) * 4];
                const float r = std::fabs(s[3]);
                const float lo[3] = {s[0] - r, s[1] - r, s[2] - r}, hi[3] = {s[0] + r, s[1] + r, s[2] + r};
                grow(b, lo);
                grow(b, hi);
            } else {
                if (i >= g.triCount) {
                    out.badRefs += 1;
                    continue;
                }
                if (triSeen[i]++) out.multiplyReferenced += 1;
                out.trianglesReferenced += 1;
                const float* t = &g.triData[static_cast<size_t>(i) * 12];
                const float v1[3] = {t[0] - t[4], t[1] - t[5], t[2] - t[6]};
                const float v2[3] = {t[0] + t[8], t[1] + t[9], t[2] + t[10]};
                grow(b, t);
                grow(b, v1);
                grow(b, v2);
            }
        }
    };

    // device indices are preorder (children after parents): depths flow forwards, bounds backwards
    for (size_t i = 0; i < nodeCount; ++i) {
        const float* n = &bvh.nodes[i * 16];
        for (int c = 0; c < 2; ++c) {
            const uint32_t ref = floatBits(n[c == 0 ? 3 : 7]);
            if (ref == ptrk::kRefEmpty) continue;
            if (ref & ptrk::kRefLeafBit) {
                out.maxDepth = std::max<uint64_t>(out.maxDepth, depth[i] + 1u);
            } else if (ref <= i || ref >= nodeCount) {
                out.badRefs += 1;
            } else {
                depth[ref] = depth[i] + 1u;
                out.maxDepth = std::max<uint64_t>(out.maxDepth, depth[ref]);
            }
        }
    }
    for (size_t ii = nodeCount; ii-- > 0;) {
        const float* n = &bvh.nodes[ii * 16];
        const uint32_t* q = bvh.qnodes.empty() ? nullptr : &bvh.qnodes[ii * 8];
        for (int c = 0; c < 2; ++c) {
            const uint32_t ref = floatBits(n[c == 0 ? 3 : 7]);
            if (ref == ptrk::kRefEmpty) continue;
            Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
            if (ref & ptrk::kRefLeafBit) {
                leafBounds(ref, b);
            } else if (ref > ii && ref < nodeCount) {
                b = subtree[ref];
            } else {
                continue;
            }
            const float* lo = n + c * 8;
            const float* hi = n + c * 8 + 4;
            bool inside = true;
            for (int a = 0; a < 3; ++a) inside = inside && b.lo[a] >= lo[a] && b.hi[a] <= hi[a];
            if (!inside) out.boxViolations += 1;
            grow(subtree[ii], b.lo);
            grow(subtree[ii], b.hi);
            if (q) {
                const uint32_t* w = q + c * 4;
                const uint32_t qlo[3] = {w[0] & 0xFFFFu, w[0] >> 16, w[1] & 0xFFFFu};
                const uint32_t qhi[3] = {w[1] >> 16, w[2] & 0xFFFFu, w[2] >> 16};
                bool ok = (w[3] == ref);
                for (int a = 0; a < 3; ++a) {
                    const double dlo = static_cast<double>(bvh.gridOrigin[a]) + static_cast<double>(qlo[a]) * bvh.gridCell[a];
                    const double dhi = static_cast<double>(bvh.gridOrigin[a]) + static_cast<double>(qhi[a]) * bvh.gridCell[a];
                    ok = ok && dlo <= lo[a] && dhi >= hi[a];
                }
                if (!ok) out.quantViolations += 1;
            }
        }
    }
    if (bvh.oversizeRef != ptrk::kRefEmpty) {
        // the triangles kept out of the tree: one leaf reference, tested by every ray before the walk
        Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
        if (!(bvh.oversizeRef & ptrk::kRefLeafBit) || (bvh.oversizeRef & ptrk::kRefSphereBit)) out.badRefs += 1;
        leafBounds(bvh.oversizeRef, b);
        out.oversize = ((bvh.oversizeRef >> ptrk::kRefCountShift) & 0xFu) + 1u;
    }
    for (uint8_t s : triSeen) out.unreferenced += (s == 0);
    for (uint8_t s : sphSeen) out.unreferenced += (s == 0);

    // the four-wide nodes of the persistent kernels: walked from the root, every primitive of the tree must turn up exactly once
    // (both ways of collapsing: by area - the default - and by level)
    for (int pass = 0; pass < 2 && nodeCount > 0 && !bvh.qnodes.empty(); ++pass) {
        std::unique_ptr<uint32_t[]> wide;
        uint32_t wideDepth = 0;
        const uint32_t wideCount = BuildWideNodes(bvh, pass == 0 ? WideCollapse::ByArea : WideCollapse::ByLevel, wide, &wideDepth);
        if (pass == 0) {
            out.wideNodes = wideCount;
            out.wideDepth = wideDepth;
        }
        std::vector<uint8_t> triWide(g.triCount, 0), sphWide(g.sphereCount, 0), visited(wideCount, 0);
        std::vector<uint32_t> stack{0u};
        while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            if (n >= wideCount || visited[n]++) {
                out.wideProblems += 1;
                continue;
            }
            for (uint32_t c = 0; c < 4u; ++c) {
                const uint32_t* rec = &wide[static_cast<size_t>(n) * 16u + c * 4u];
                const uint32_t ref = rec[3];
                // a used place holds an ordered box, an unused one the inverted box the wide step relies on (it does not read the reference)
                const bool inverted = (rec[0] & 0xFFFFu) > (rec[1] >> 16) && (rec[0] >> 16) > (rec[2] & 0xFFFFu) && (rec[1] & 0xFFFFu) > (rec[2] >> 16);
                const bool ordered = (rec[0] & 0xFFFFu) <= (rec[1] >> 16) && (rec[0] >> 16) <= (rec[2] & 0xFFFFu) && (rec[1] & 0xFFFFu) <= (rec[2] >> 16);
                if ((ref == ptrk::kRefEmpty) ? !inverted : !ordered) out.wideProblems += 1;
                if (ref == ptrk::kRefEmpty) continue;
                if (!(ref & ptrk::kRefLeafBit)) {
                    stack.push_back(ref);
                    continue;
                }
                const uint32_t first = ref & ptrk::kRefOffsetMask, count = ((ref >> ptrk::kRefCountShift) & 0xFu) + 1u;
                std::vector<uint8_t>& seen = (ref & ptrk::kRefSphereBit) ? sphWide : triWide;
                for (uint32_t i = first; i < first + count; ++i) {
                    if (i >= seen.size() || seen[i]++) out.wideProblems += 1;
                }
            }
        }
        for (uint32_t n = 0; n < wideCount; ++n) out.wideProblems += (visited[n] == 0);
        const uint32_t outside = bvh.oversizeRef != ptrk::kRefEmpty ? ((bvh.oversizeRef >> ptrk::kRefCountShift) & 0xFu) + 1u : 0u;
        uint64_t missed = 0;
        for (uint8_t s : triWide) missed += (s == 0);
        for (uint8_t s : sphWide) missed += (s == 0);
        if (missed != outside) out.wideProblems += 1 + (missed > outside ? missed - outside : outside - missed);
    }
}

}  // namespace ptr
