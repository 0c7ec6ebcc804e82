// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md and oracle_raycast.h.
#include "oracle_raycast.h"

#include <cstring>
#include <limits>
#include <numeric>

namespace oracle {
namespace {

struct M4 {
    float m[4][4];  // m[col][row], column-major like simd::float4x4
};

M4 loadM4(const float* colMajor) {
    M4 r;
    std::memcpy(r.m, colMajor, sizeof(r.m));
    return r;
}

// General 4x4 inverse by cofactors (stands in for simd_inverse, EmbreeHeadlessRenderer.mm:2057).
M4 inverse(const M4& a) {
    const float* s = &a.m[0][0];
    float inv[16];
    inv[0] = s[5] * s[10] * s[15] - s[5] * s[11] * s[14] - s[9] * s[6] * s[15] + s[9] * s[7] * s[14] + s[13] * s[6] * s[11] - s[13] * s[7] * s[10];
    inv[4] = -s[4] * s[10] * s[15] + s[4] * s[11] * s[14] + s[8] * s[6] * s[15] - s[8] * s[7] * s[14] - s[12] * s[6] * s[11] + s[12] * s[7] * s[10];
    inv[8] = s[4] * s[9] * s[15] - s[4] * s[11] * s[13] - s[8] * s[5] * s[15] + s[8] * s[7] * s[13] + s[12] * s[5] * s[11] - s[12] * s[7] * s[9];
    inv[12] = -s[4] * s[9] * s[14] + s[4] * s[10] * s[13] + s[8] * s[5] * s[14] - s[8] * s[6] * s[13] - s[12] * s[5] * s[10] + s[12] * s[6] * s[9];
    inv[1] = -s[1] * s[10] * s[15] + s[1] * s[11] * s[14] + s[9] * s[2] * s[15] - s[9] * s[3] * s[14] - s[13] * s[2] * s[11] + s[13] * s[3] * s[10];
    inv[5] = s[0] * s[10] * s[15] - s[0] * s[11] * s[14] - s[8] * s[2] * s[15] + s[8] * s[3] * s[14] + s[12] * s[2] * s[11] - s[12] * s[3] * s[10];
    inv[9] = -s[0] * s[9] * s[15] + s[0] * s[11] * s[13] + s[8] * s[1] * s[15] - s[8] * s[3] * s[13] - s[12] * s[1] * s[11] + s[12] * s[3] * s[9];
    inv[13] = s[0] * s[9] * s[14] - s[0] * s[10] * s[13] - s[8] * s[1] * s[14] + s[8] * s[2] * s[13] + s[12] * s[1] * s[10] - s[12] * s[2] * s[9];
    inv[2] = s[1] * s[6] * s[15] - s[1] * s[7] * s[14] - s[5] * s[2] * s[15] + s[5] * s[3] * s[14] + s[13] * s[2] * s[7] - s[13] * s[3] * s[6];
    inv[6] = -s[0] * s[6] * s[15] + s[0] * s[7] * s[14] + s[4] * s[2] * s[15] - s[4] * s[3] * s[14] - s[12] * s[2] * s[7] + s[12] * s[3] * s[6];
    inv[10] = s[0] * s[5] * s[15] - s[0] * s[7] * s[13] - s[4] * s[1] * s[15] + s[4] * s[3] * s[13] + s[12] * s[1] * s[7] - s[12] * s[3] * s[5];
    inv[14] = -s[0] * s[5] * s[14] + s[0] * s[6] * s[13] + s[4] * s[1] * s[14] - s[4] * s[2] * s[13] - s[12] * s[1] * s[6] + s[12] * s[2] * s[5];
    inv[3] = -s[1] * s[6] * s[11] + s[1] * s[7] * s[10] + s[5] * s[2] * s[11] - s[5] * s[3] * s[10] - s[9] * s[2] * s[7] + s[9] * s[3] * s[6];
    inv[7] = s[0] * s[6] * s[11] - s[0] * s[7] * s[10] - s[4] * s[2] * s[11] + s[4] * s[3] * s[10] + s[8] * s[2] * s[7] - s[8] * s[3] * s[6];
    inv[11] = -s[0] * s[5] * s[11] + s[0] * s[7] * s[9] + s[4] * s[1] * s[11] - s[4] * s[3] * s[9] - s[8] * s[1] * s[7] + s[8] * s[3] * s[5];
    inv[15] = s[0] * s[5] * s[10] - s[0] * s[6] * s[9] - s[4] * s[1] * s[10] + s[4] * s[2] * s[9] + s[8] * s[1] * s[6] - s[8] * s[2] * s[5];
    float det = s[0] * inv[0] + s[1] * inv[4] + s[2] * inv[8] + s[3] * inv[12];
    M4 r;
    const float invDet = 1.0f / det;
    for (int i = 0; i < 16; ++i) (&r.m[0][0])[i] = inv[i] * invDet;
    return r;
}

// TransformPoint, EmbreeHeadlessRenderer.mm:2051-2054
V3 transformPoint(const M4& t, V3 p) {
    return {((t.m[0][0] * p.x + t.m[1][0] * p.y) + t.m[2][0] * p.z) + t.m[3][0],
            ((t.m[0][1] * p.x + t.m[1][1] * p.y) + t.m[2][1] * p.z) + t.m[3][1],
            ((t.m[0][2] * p.x + t.m[1][2] * p.y) + t.m[2][2] * p.z) + t.m[3][2]};
}

Prim makeTriangle(V3 v0, V3 v1, V3 v2, uint32_t geom, uint32_t primId) {
    Prim p;
    p.v0 = v0;
    p.e1 = v0 - v1;
    p.e2 = v2 - v0;
    p.geom = geom;
    p.primId = primId;
    p.isSphere = 0;
    return p;
}

struct Box {
    float lo[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
    float hi[3] = {-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity()};
    void grow(const float p[3]) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float halfArea() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

Box primBox(const Prim& p) {
    Box b;
    if (p.isSphere) {
        const float r = std::fabs(p.e1.x);
        const float lo[3] = {p.v0.x - r, p.v0.y - r, p.v0.z - r};
        const float hi[3] = {p.v0.x + r, p.v0.y + r, p.v0.z + r};
        b.grow(lo);
        b.grow(hi);
    } else {
        const V3 v1 = p.v0 - p.e1, v2 = p.v0 + p.e2;
        const float a[3] = {p.v0.x, p.v0.y, p.v0.z}, bb[3] = {v1.x, v1.y, v1.z}, c[3] = {v2.x, v2.y, v2.z};
        b.grow(a);
        b.grow(bb);
        b.grow(c);
    }
    // pad so that slab rounding can never cull a primitive the exact test would accept
    for (int a = 0; a < 3; ++a) {
        const float pad = 1e-5f * std::max(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), 1.0f);
        b.lo[a] -= pad;
        b.hi[a] += pad;
    }
    return b;
}

inline bool testTriangle(const Prim& p, V3 org, V3 dir, float tnear, float tfar, float& t, float& u, float& v, V3& ng) {
    const V3 Ng = cross(p.e2, p.e1);
    const V3 C = p.v0 - org;
    const V3 R = cross(C, dir);
    const float den = dot(Ng, dir);
    const float absDen = std::fabs(den);
    const float sgn = std::signbit(den) ? -1.0f : 1.0f;
    const float U = dot(R, p.e2) * sgn;
    const float V = dot(R, p.e1) * sgn;
    if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && (U + V) <= absDen)) return false;
    const float T = dot(Ng, C) * sgn;
    if (!(absDen * tnear < T && T <= absDen * tfar)) return false;
    const float rcpAbsDen = 1.0f / absDen;
    t = T * rcpAbsDen;
    u = U * rcpAbsDen;
    v = V * rcpAbsDen;
    ng = Ng;
    return true;
}

inline bool testSphere(const Prim& p, V3 org, V3 dir, float tnear, float tfar, float& t, V3& ng) {
    const float rd2 = 1.0f / dot(dir, dir);
    const V3 c0 = p.v0 - org;
    const float projC0 = dot(c0, dir) * rd2;
    const V3 perp = c0 - projC0 * dir;
    const float l2 = dot(perp, perp);
    const float r2 = p.e1.x * p.e1.x;
    if (!(l2 <= r2)) return false;
    const float td = std::sqrt((r2 - l2) * rd2);
    const float tFront = projC0 - td;
    const float tBack = projC0 + td;
    const bool validFront = (tnear <= tFront) && (tFront <= tfar);
    const bool validBack = (tnear <= tBack) && (tBack <= tfar);
    if (!validFront && !validBack) return false;
    t = validFront ? tFront : tBack;
    ng = (validFront ? -td : td) * dir - perp;
    return true;
}

}  // namespace

void Scene::build(const PtrSceneDesc& desc) {
    geoms.clear();
    prims.clear();
    nodes.clear();
    spheres.clear();
    rects = desc.rects;
    rectCount = desc.rectCount;

    for (uint32_t mi = 0; mi < desc.meshCount; ++mi) {
        const PtrMeshDesc& mesh = desc.meshes[mi];
        if (mesh.vertexCount == 0 || mesh.indexCount == 0) continue;
        const M4 l2w = loadM4(mesh.localToWorld);
        const M4 w2l = inverse(l2w);
        // NormalMatrix: columns are the rows of worldToLocal's upper 3x3 (EmbreeHeadlessRenderer.mm:2056-2069)
        const V3 nc0{w2l.m[0][0], w2l.m[1][0], w2l.m[2][0]};
        const V3 nc1{w2l.m[0][1], w2l.m[1][1], w2l.m[2][1]};
        const V3 nc2{w2l.m[0][2], w2l.m[1][2], w2l.m[2][2]};

        Geom g;
        g.type = GeomType::Mesh;
        g.meshIndex = mi;
        g.materialIndex = mesh.materialIndex;
        std::vector<V3> positions(mesh.vertexCount);
        g.normals.resize(mesh.vertexCount);
        for (uint32_t v = 0; v < mesh.vertexCount; ++v) {
            positions[v] = transformPoint(l2w, V3(mesh.positions + 3 * v));
            const V3 n(mesh.normals + 3 * v);
            const V3 wn = (nc0 * n.x + nc1 * n.y) + nc2 * n.z;
            g.normals[v] = length(wn) > 0.0f ? normalize(wn) : wn;
        }
        g.indices.assign(mesh.indices, mesh.indices + mesh.indexCount);
        // texture attributes (shaders/pathtrace.metal:640-739): world-space tangents through the linear part of localToWorld
        const float det3 = l2w.m[0][0] * (l2w.m[1][1] * l2w.m[2][2] - l2w.m[2][1] * l2w.m[1][2]) -
                           l2w.m[1][0] * (l2w.m[0][1] * l2w.m[2][2] - l2w.m[2][1] * l2w.m[0][2]) +
                           l2w.m[2][0] * (l2w.m[0][1] * l2w.m[1][2] - l2w.m[1][1] * l2w.m[0][2]);
        g.detSign = det3 < 0.0f ? -1.0f : 1.0f;
        if (mesh.uv0 || mesh.uv1) {
            g.positions = positions;
            g.uv0.assign(static_cast<size_t>(mesh.vertexCount) * 2, 0.0f);
            g.uv1.assign(static_cast<size_t>(mesh.vertexCount) * 2, 0.0f);
            if (mesh.uv0) g.uv0.assign(mesh.uv0, mesh.uv0 + static_cast<size_t>(mesh.vertexCount) * 2);
            if (mesh.uv1) g.uv1.assign(mesh.uv1, mesh.uv1 + static_cast<size_t>(mesh.vertexCount) * 2);
            if (mesh.tangents) {
                g.tangents.resize(static_cast<size_t>(mesh.vertexCount) * 4);
                for (uint32_t v = 0; v < mesh.vertexCount; ++v) {
                    const float* tl = mesh.tangents + 4 * v;
                    g.tangents[4 * v + 0] = (l2w.m[0][0] * tl[0] + l2w.m[1][0] * tl[1]) + l2w.m[2][0] * tl[2];
                    g.tangents[4 * v + 1] = (l2w.m[0][1] * tl[0] + l2w.m[1][1] * tl[1]) + l2w.m[2][1] * tl[2];
                    g.tangents[4 * v + 2] = (l2w.m[0][2] * tl[0] + l2w.m[1][2] * tl[1]) + l2w.m[2][2] * tl[2];
                    g.tangents[4 * v + 3] = tl[3] == 0.0f ? 0.0f : (tl[3] < 0.0f ? -1.0f : 1.0f) * g.detSign;
                }
            }
        }
        const uint32_t geomId = static_cast<uint32_t>(geoms.size());
        for (uint32_t t = 0; t + 2 < mesh.indexCount; t += 3) {
            prims.push_back(makeTriangle(positions[mesh.indices[t]], positions[mesh.indices[t + 1]],
                                         positions[mesh.indices[t + 2]], geomId, t / 3));
        }
        geoms.push_back(std::move(g));
    }

    if (desc.sphereCount > 0 && desc.spheres) {
        Geom g;
        g.type = GeomType::Spheres;
        const uint32_t geomId = static_cast<uint32_t>(geoms.size());
        for (uint32_t i = 0; i < desc.sphereCount; ++i) {
            spheres.push_back(desc.spheres[i]);
            g.primMaterial.push_back(desc.spheres[i].materialIndex[0]);
            Prim p;
            p.v0 = V3(desc.spheres[i].centerRadius);
            p.e1 = V3(desc.spheres[i].centerRadius[3], 0.0f, 0.0f);
            p.e2 = V3();
            p.geom = geomId;
            p.primId = i;
            p.isSphere = 1;
            prims.push_back(p);
        }
        geoms.push_back(std::move(g));
    }

    if (desc.rectCount > 0 && desc.rects) {
        // two triangles per rectangle, winding chosen so the geometric normal agrees with the stored one
        // (EmbreeHeadlessRenderer.mm:2211-2257)
        Geom g;
        g.type = GeomType::Rectangles;
        const uint32_t geomId = static_cast<uint32_t>(geoms.size());
        std::vector<V3> positions;
        for (uint32_t i = 0; i < desc.rectCount; ++i) {
            const PtrRect& r = desc.rects[i];
            const V3 corner(r.corner), eu(r.edgeU), ev(r.edgeV);
            const V3 normal = normalize(V3(r.normalAndPlane));
            const uint32_t base = static_cast<uint32_t>(positions.size());
            positions.push_back(corner);
            positions.push_back(corner + eu);
            positions.push_back(corner + ev);
            positions.push_back((corner + eu) + ev);
            for (int j = 0; j < 4; ++j) g.normals.push_back(normal);
            const bool flip = dot(normalize(cross(eu, ev)), normal) < 0.0f;
            const uint32_t order[2][6] = {{0, 1, 2, 2, 1, 3}, {0, 2, 1, 1, 2, 3}};
            for (int k = 0; k < 6; ++k) g.indices.push_back(base + order[flip ? 1 : 0][k]);
            for (int k = 0; k < 2; ++k) {
                g.primMaterial.push_back(r.materialTwoSided[0]);
                g.triToRect.push_back(i);
            }
        }
        for (uint32_t t = 0; t + 2 < g.indices.size(); t += 3) {
            prims.push_back(makeTriangle(positions[g.indices[t]], positions[g.indices[t + 1]],
                                         positions[g.indices[t + 2]], geomId, t / 3));
        }
        geoms.push_back(std::move(g));
    }
    buildBvh();
}

void Scene::buildBvh() {
    const uint32_t n = static_cast<uint32_t>(prims.size());
    nodes.clear();
    if (n == 0) return;
    std::vector<Box> boxes(n);
    std::vector<V3> centers(n);
    for (uint32_t i = 0; i < n; ++i) {
        boxes[i] = primBox(prims[i]);
        centers[i] = V3(0.5f * (boxes[i].lo[0] + boxes[i].hi[0]), 0.5f * (boxes[i].lo[1] + boxes[i].hi[1]),
                        0.5f * (boxes[i].lo[2] + boxes[i].hi[2]));
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    nodes.reserve(2 * n);
    nodes.push_back(BvhNode{});

    struct Work {
        uint32_t node, begin, end;
    };
    std::vector<Work> stack{{0, 0, n}};
    constexpr int kBins = 16;
    constexpr uint32_t kMaxLeaf = 4;
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        Box bounds, cbounds;
        for (uint32_t i = w.begin; i < w.end; ++i) {
            bounds.grow(boxes[order[i]]);
            const float c[3] = {centers[order[i]].x, centers[order[i]].y, centers[order[i]].z};
            cbounds.grow(c);
        }
        BvhNode& node = nodes[w.node];
        std::memcpy(node.lo, bounds.lo, sizeof(node.lo));
        std::memcpy(node.hi, bounds.hi, sizeof(node.hi));
        const uint32_t count = w.end - w.begin;
        auto makeLeaf = [&]() {
            nodes[w.node].left = w.begin;
            nodes[w.node].count = count;
        };
        if (count <= 2) {
            makeLeaf();
            continue;
        }
        int bestAxis = -1, bestBin = -1;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int axis = 0; axis < 3; ++axis) {
            const float lo = cbounds.lo[axis], hi = cbounds.hi[axis];
            if (!(hi > lo)) continue;
            Box binBox[kBins];
            uint32_t binCount[kBins] = {0};
            const float scale = kBins / (hi - lo);
            for (uint32_t i = w.begin; i < w.end; ++i) {
                const float c = (&centers[order[i]].x)[axis];
                const int b = std::min(kBins - 1, static_cast<int>((c - lo) * scale));
                binBox[b].grow(boxes[order[i]]);
                ++binCount[b];
            }
            float rightArea[kBins];
            uint32_t rightCount[kBins];
            Box acc;
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(binBox[b]);
                cnt += binCount[b];
                rightArea[b] = acc.halfArea();
                rightCount[b] = cnt;
            }
            acc = Box();
            cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(binBox[b]);
                cnt += binCount[b];
                if (cnt == 0 || rightCount[b + 1] == 0) continue;
                const float cost = acc.halfArea() * cnt + rightArea[b + 1] * rightCount[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = b;
                }
            }
        }
        uint32_t mid = w.begin;
        if (bestAxis >= 0) {
            const float leafCost = bounds.halfArea() * count;
            if (count <= kMaxLeaf && bestCost >= leafCost) {
                makeLeaf();
                continue;
            }
            const float lo = cbounds.lo[bestAxis], hi = cbounds.hi[bestAxis];
            const float scale = kBins / (hi - lo);
            auto it = std::partition(order.begin() + w.begin, order.begin() + w.end, [&](uint32_t id) {
                const float c = (&centers[id].x)[bestAxis];
                return std::min(kBins - 1, static_cast<int>((c - lo) * scale)) <= bestBin;
            });
            mid = static_cast<uint32_t>(it - order.begin());
        }
        if (mid == w.begin || mid == w.end) {
            if (count <= kMaxLeaf) {
                makeLeaf();
                continue;
            }
            mid = w.begin + count / 2;  // coincident centroids: split by index
        }
        const uint32_t left = static_cast<uint32_t>(nodes.size());
        nodes.push_back(BvhNode{});
        nodes.push_back(BvhNode{});
        nodes[w.node].left = left;
        nodes[w.node].count = 0;
        stack.push_back({left + 1, mid, w.end});
        stack.push_back({left, w.begin, mid});
    }
    std::vector<Prim> sorted(n);
    for (uint32_t i = 0; i < n; ++i) sorted[i] = prims[order[i]];
    prims.swap(sorted);
}

namespace {

inline bool slab(const BvhNode& nd, V3 org, V3 inv, float tnear, float tfar, float& entry) {
    float t0 = tnear, t1 = tfar;
    const float o[3] = {org.x, org.y, org.z}, iv[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
        float ta = (nd.lo[a] - o[a]) * iv[a];
        float tb = (nd.hi[a] - o[a]) * iv[a];
        if (ta > tb) std::swap(ta, tb);
        // NaN (0 * inf) compares false and leaves the interval untouched
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    entry = t0;
    return t0 <= t1 * 1.0000004f;
}

}  // namespace

bool Scene::intersect(V3 org, V3 dir, float tnear, float tfar, RayHit& hit, bool bruteForce, Counters* counters) const {
    hit = RayHit{};
    bool found = false;
    auto testPrim = [&](const Prim& p) {
        if (counters) ++counters->prims;
        float t, u = 0.0f, v = 0.0f;
        V3 ng;
        const bool ok = p.isSphere ? testSphere(p, org, dir, tnear, tfar, t, ng) : testTriangle(p, org, dir, tnear, tfar, t, u, v, ng);
        if (ok) {
            tfar = t;
            hit.t = t;
            hit.u = u;
            hit.v = v;
            hit.ng = ng;
            hit.geom = p.geom;
            hit.primId = p.primId;
            found = true;
        }
    };
    if (bruteForce || nodes.empty()) {
        for (const Prim& p : prims) testPrim(p);
        return found;
    }
    const V3 inv{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    struct Entry {
        uint32_t node;
        float entry;
    };
    Entry stack[128];
    int sp = 0;
    float e0;
    if (!slab(nodes[0], org, inv, tnear, tfar, e0)) return false;
    stack[sp++] = {0, e0};
    while (sp > 0) {
        const Entry cur = stack[--sp];
        if (cur.entry > tfar) continue;
        const BvhNode& nd = nodes[cur.node];
        if (counters) ++counters->nodes;
        if (nd.count > 0) {
            for (uint32_t i = 0; i < nd.count; ++i) testPrim(prims[nd.left + i]);
            continue;
        }
        float el, er;
        const bool hl = slab(nodes[nd.left], org, inv, tnear, tfar, el);
        const bool hr = slab(nodes[nd.left + 1], org, inv, tnear, tfar, er);
        if (hl && hr) {
            if (el <= er) {
                stack[sp++] = {nd.left + 1, er};
                stack[sp++] = {nd.left, el};
            } else {
                stack[sp++] = {nd.left, el};
                stack[sp++] = {nd.left + 1, er};
            }
        } else if (hl) {
            stack[sp++] = {nd.left, el};
        } else if (hr) {
            stack[sp++] = {nd.left + 1, er};
        }
    }
    return found;
}

bool Scene::occluded(V3 org, V3 dir, float tnear, float tfar, bool bruteForce, Counters* counters) const {
    auto testPrim = [&](const Prim& p) {
        if (counters) ++counters->prims;
        float t, u, v;
        V3 ng;
        return p.isSphere ? testSphere(p, org, dir, tnear, tfar, t, ng) : testTriangle(p, org, dir, tnear, tfar, t, u, v, ng);
    };
    if (bruteForce || nodes.empty()) {
        for (const Prim& p : prims) {
            if (testPrim(p)) return true;
        }
        return false;
    }
    const V3 inv{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    uint32_t stack[128];
    int sp = 0;
    float e;
    if (!slab(nodes[0], org, inv, tnear, tfar, e)) return false;
    stack[sp++] = 0;
    while (sp > 0) {
        const BvhNode& nd = nodes[stack[--sp]];
        if (counters) ++counters->nodes;
        if (nd.count > 0) {
            for (uint32_t i = 0; i < nd.count; ++i) {
                if (testPrim(prims[nd.left + i])) return true;
            }
            continue;
        }
        if (slab(nodes[nd.left + 1], org, inv, tnear, tfar, e)) stack[sp++] = nd.left + 1;
        if (slab(nodes[nd.left], org, inv, tnear, tfar, e)) stack[sp++] = nd.left;
    }
    return false;
}

}  // namespace oracle
