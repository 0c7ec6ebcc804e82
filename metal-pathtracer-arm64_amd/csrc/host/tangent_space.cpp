// Tangent frames by the MikkTSpace method - own implementation for triangle soups.
//
// The reference generates tangents for glTF primitives that carry none by handing the de-indexed mesh to mikktspace.c's
// genTangSpaceDefault (src/assets/TangentGen.mm:181-230, called from src/assets/GltfLoader.mm:1449).  The method (M. Mikkelsen,
// "Simulation of Wrinkled Surfaces Revisited", 2008; the de-facto standard that normal-map bakers assume):
//   1. corners with identical position, normal and texture coordinate are one vertex (welding);
//   2. a triangle with two coinciding positions is degenerate and takes no part; the rest get their first-order derivatives
//      dP/ds, dP/dt (unit length, flipped where the mapping mirrors), their magnitudes, and whether the mapping preserves orientation;
//   3. triangles are neighbours across an edge that both run in opposite directions;
//   4. around every vertex, the triangles that can be reached through neighbours sharing the vertex and agree on orientation form
//      a group (a triangle whose texture mapping is degenerate joins whichever group reaches it first);
//   5. a corner's tangent is the angle-weighted sum of the group's derivatives projected into the plane of the vertex normal
//      (members whose projected derivatives differ by more than the angular threshold - 180 degrees by default - form sub-groups);
//   6. corners of degenerate triangles copy the frame of a good triangle at the same vertex.
// Sums run over members in ascending triangle order and every product, sum, division and square root is a single-precision
// operation in the order the method states them, so the frames agree with the reference library's (tests/test_reference_loaders.py
// holds this file to mikktspace.c compiled from the reference tree: tangent and sign per corner).
#include "tangent_space.h"
#include "vecmath.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace ptr {

namespace {

struct V3 {
    float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 scale(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(V3 v) { return sqrtf(dot(v, v)); }
inline V3 unit(V3 v) { return scale(1.0f / length(v), v); }
inline bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool notZero(float x) { return fabsf(x) > FLT_MIN; }
inline bool notZero(V3 v) { return notZero(v.x) || notZero(v.y) || notZero(v.z); }
// a derivative projected into the plane of normal n, unit length when it has any
inline V3 projected(V3 n, V3 v) {
    V3 p = sub(v, scale(dot(n, v), n));
    if (notZero(p)) p = unit(p);
    return p;
}

struct Frame {
    V3 os{1.0f, 0.0f, 0.0f}, ot{0.0f, 1.0f, 0.0f};
    float magS = 1.0f, magT = 1.0f;
    bool orient = false;
};

struct Tri {
    uint32_t source = 0;        // triangle of the input
    uint32_t v[3] = {0, 0, 0};  // welded vertex of each corner (= index of a representative corner of the input)
    int32_t neighbour[3] = {-1, -1, -1};   // across edge (corner i, corner i + 1)
    int32_t group[3] = {-1, -1, -1};
    V3 os{0.0f, 0.0f, 0.0f}, ot{0.0f, 0.0f, 0.0f};
    float magS = 0.0f, magT = 0.0f;
    bool orientPreserving = false, groupWithAny = true;
};

struct Group {
    uint32_t vertex = 0;
    bool orientPreserving = false;
    std::vector<uint32_t> faces;
};

struct Mesh {
    const float* pos;
    const float* nrm;
    const float* uv;
    V3 position(uint32_t corner) const { return {pos[3 * corner], pos[3 * corner + 1], pos[3 * corner + 2]}; }
    V3 normal(uint32_t corner) const { return {nrm[3 * corner], nrm[3 * corner + 1], nrm[3 * corner + 2]}; }
    float s(uint32_t corner) const { return uv[2 * corner]; }
    float t(uint32_t corner) const { return uv[2 * corner + 1]; }
};

// ---- 1. welding: corners that compare equal in all eight floats ----
struct Key {
    uint32_t w[8];
    bool operator==(const Key& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 0xcbf29ce484222325ull;
        for (uint32_t x : k.w) {
            h ^= x;
            h *= 0x100000001b3ull;
        }
        return static_cast<size_t>(h);
    }
};

void weld(const Mesh& m, size_t corners, std::vector<uint32_t>& vertexOf) {
    vertexOf.resize(corners);
    std::unordered_map<Key, uint32_t, KeyHash> seen;
    seen.reserve(corners);
    for (size_t c = 0; c < corners; ++c) {
        const float f[8] = {m.pos[3 * c], m.pos[3 * c + 1], m.pos[3 * c + 2], m.nrm[3 * c], m.nrm[3 * c + 1], m.nrm[3 * c + 2], m.uv[2 * c], m.uv[2 * c + 1]};
        Key k;
        bool comparable = true;
        for (int i = 0; i < 8; ++i) {
            const float value = f[i] == 0.0f ? 0.0f : f[i];   // -0 equals +0
            comparable = comparable && value == value;        // a NaN equals nothing, itself included
            std::memcpy(&k.w[i], &value, 4);
        }
        if (!comparable) {
            vertexOf[c] = static_cast<uint32_t>(c);
            continue;
        }
        vertexOf[c] = seen.try_emplace(k, static_cast<uint32_t>(c)).first->second;
    }
}

// ---- 3. neighbours ----
struct Edge {
    uint32_t lo, hi, face;
};

void edgeOf(const Tri& t, uint32_t lo, uint32_t hi, uint32_t& from, uint32_t& to, int& number) {
    // the edge of t whose two vertices are lo / hi, in the direction t runs it
    for (int i = 0; i < 3; ++i) {
        const uint32_t a = t.v[i], b = t.v[(i + 1) % 3];
        if ((a == lo && b == hi) || (a == hi && b == lo)) {
            from = a;
            to = b;
            number = i;
            return;
        }
    }
    from = to = 0;
    number = 0;
}

void findNeighbours(std::vector<Tri>& tris) {
    std::vector<Edge> edges;
    edges.reserve(tris.size() * 3);
    for (uint32_t f = 0; f < tris.size(); ++f) {
        for (int i = 0; i < 3; ++i) {
            const uint32_t a = tris[f].v[i], b = tris[f].v[(i + 1) % 3];
            edges.push_back({std::min(a, b), std::max(a, b), f});
        }
    }
    std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) {
        if (a.lo != b.lo) return a.lo < b.lo;
        if (a.hi != b.hi) return a.hi < b.hi;
        return a.face < b.face;
    });
    // an unmatched edge pairs with the next unmatched edge over the same two vertices that runs the other way (where more than two
    // triangles meet in an edge, lower triangle numbers pair first)
    for (size_t i = 0; i < edges.size(); ++i) {
        const Edge& e = edges[i];
        uint32_t fromA, toA;
        int numberA;
        edgeOf(tris[e.face], e.lo, e.hi, fromA, toA, numberA);
        if (tris[e.face].neighbour[numberA] != -1) continue;
        for (size_t j = i + 1; j < edges.size() && edges[j].lo == e.lo && edges[j].hi == e.hi; ++j) {
            uint32_t fromB, toB;
            int numberB;
            edgeOf(tris[edges[j].face], e.lo, e.hi, fromB, toB, numberB);
            if (fromA == toB && toA == fromB && tris[edges[j].face].neighbour[numberB] == -1) {
                tris[e.face].neighbour[numberA] = static_cast<int32_t>(edges[j].face);
                tris[edges[j].face].neighbour[numberB] = static_cast<int32_t>(e.face);
                break;
            }
        }
    }
}

// ---- 4. groups ----
void growGroup(std::vector<Tri>& tris, Group& g, int32_t groupIndex, int32_t start) {
    // depth first, the neighbour across the edge leaving the vertex before the one across the edge arriving at it
    std::vector<int32_t> todo{start};
    while (!todo.empty()) {
        const int32_t f = todo.back();
        todo.pop_back();
        Tri& t = tris[static_cast<size_t>(f)];
        int i = t.v[0] == g.vertex ? 0 : (t.v[1] == g.vertex ? 1 : (t.v[2] == g.vertex ? 2 : -1));
        if (i < 0 || t.group[i] != -1) continue;   // (already in this group, or in another)
        if (t.groupWithAny && t.group[0] == -1 && t.group[1] == -1 && t.group[2] == -1) {
            t.orientPreserving = g.orientPreserving;   // the first group to reach a triangle without a usable mapping decides its side
        }
        if (t.orientPreserving != g.orientPreserving) continue;
        g.faces.push_back(static_cast<uint32_t>(f));
        t.group[i] = groupIndex;
        const int32_t left = t.neighbour[i], right = t.neighbour[i > 0 ? i - 1 : 2];
        if (right >= 0) todo.push_back(right);
        if (left >= 0) todo.push_back(left);
    }
}

// ---- 5. the frame of a set of triangles around vertex `vertex` ----
Frame evaluate(const Mesh& m, const std::vector<Tri>& tris, const std::vector<uint32_t>& members, uint32_t vertex) {
    Frame res;
    res.os = res.ot = {0.0f, 0.0f, 0.0f};
    res.magS = res.magT = 0.0f;
    float angleSum = 0.0f;
    for (uint32_t f : members) {
        const Tri& t = tris[f];
        if (t.groupWithAny) continue;   // only triangles with a usable mapping contribute
        const int i = t.v[0] == vertex ? 0 : (t.v[1] == vertex ? 1 : 2);
        const V3 n = m.normal(t.v[i]);
        const V3 os = projected(n, t.os), ot = projected(n, t.ot);
        const V3 p0 = m.position(t.v[i > 0 ? i - 1 : 2]), p1 = m.position(t.v[i]), p2 = m.position(t.v[i < 2 ? i + 1 : 0]);
        const V3 e1 = projected(n, sub(p0, p1)), e2 = projected(n, sub(p2, p1));
        float c = dot(e1, e2);
        c = c > 1.0f ? 1.0f : (c < -1.0f ? -1.0f : c);
        const float angle = static_cast<float>(acos(static_cast<double>(c)));
        res.os = add(res.os, scale(angle, os));
        res.ot = add(res.ot, scale(angle, ot));
        res.magS += angle * t.magS;
        res.magT += angle * t.magT;
        angleSum += angle;
    }
    if (notZero(res.os)) res.os = unit(res.os);
    if (notZero(res.ot)) res.ot = unit(res.ot);
    if (angleSum > 0.0f) {
        res.magS /= angleSum;
        res.magT /= angleSum;
    }
    return res;
}

}  // namespace

bool GenerateTangentSpace(const float* positions, const float* normals, const float* uvs, size_t triangleCount, float* tangents, float angularThresholdDegrees) {
    if (triangleCount == 0 || !positions || !normals || !uvs || !tangents) return false;
    if (triangleCount > 0x2AAAAAAAu) return false;   // corner indices are 32 bit
    const Mesh m{positions, normals, uvs};
    const size_t corners = triangleCount * 3;
    const float thresholdCos = static_cast<float>(cos((angularThresholdDegrees * static_cast<float>(3.1415926535897932384626433832795)) / 180.0f));

    std::vector<uint32_t> vertexOf;
    weld(m, corners, vertexOf);

    // ---- 2. good triangles first (in their order), then the degenerate ones ----
    std::vector<Tri> tris;
    tris.reserve(triangleCount);
    std::vector<uint32_t> degenerate;
    for (uint32_t f = 0; f < triangleCount; ++f) {
        const uint32_t a = vertexOf[3 * f], b = vertexOf[3 * f + 1], c = vertexOf[3 * f + 2];
        const V3 p0 = m.position(a), p1 = m.position(b), p2 = m.position(c);
        if (same(p0, p1) || same(p0, p2) || same(p1, p2)) {
            degenerate.push_back(f);
            continue;
        }
        Tri t;
        t.source = f;
        t.v[0] = a;
        t.v[1] = b;
        t.v[2] = c;
        tris.push_back(t);
    }
    for (Tri& t : tris) {
        const V3 p0 = m.position(t.v[0]), p1 = m.position(t.v[1]), p2 = m.position(t.v[2]);
        const float s10 = m.s(t.v[1]) - m.s(t.v[0]), t10 = m.t(t.v[1]) - m.t(t.v[0]);
        const float s20 = m.s(t.v[2]) - m.s(t.v[0]), t20 = m.t(t.v[2]) - m.t(t.v[0]);
        const V3 d1 = sub(p1, p0), d2 = sub(p2, p0);
        const float signedArea2 = s10 * t20 - t10 * s20;   // twice the signed area in texture space
        const V3 os = sub(scale(t20, d1), scale(t10, d2));
        const V3 ot = add(scale(-s20, d1), scale(s10, d2));
        t.orientPreserving = signedArea2 > 0.0f;
        if (notZero(signedArea2)) {
            const float absArea = fabsf(signedArea2), lenS = length(os), lenT = length(ot);
            const float side = t.orientPreserving ? 1.0f : -1.0f;
            if (notZero(lenS)) t.os = scale(side / lenS, os);
            if (notZero(lenT)) t.ot = scale(side / lenT, ot);
            t.magS = lenS / absArea;   // magnitudes before the derivatives were made unit length
            t.magT = lenT / absArea;
            if (notZero(t.magS) && notZero(t.magT)) t.groupWithAny = false;
        }
    }
    findNeighbours(tris);

    std::vector<Group> groups;
    for (uint32_t f = 0; f < tris.size(); ++f) {
        for (int i = 0; i < 3; ++i) {
            if (tris[f].groupWithAny || tris[f].group[i] != -1) continue;
            groups.emplace_back();
            Group& g = groups.back();
            g.vertex = tris[f].v[i];
            g.orientPreserving = tris[f].orientPreserving;
            growGroup(tris, g, static_cast<int32_t>(groups.size() - 1), static_cast<int32_t>(f));
        }
    }

    // ---- 5. frames, one per (group, sub-group) ----
    std::vector<Frame> frames(corners);   // by input corner; corners nobody reaches keep the default frame
    std::vector<std::vector<uint32_t>> subGroups;
    std::vector<Frame> subFrames;
    std::vector<uint32_t> members;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group& g = groups[gi];
        subGroups.clear();
        subFrames.clear();
        for (uint32_t f : g.faces) {
            const Tri& t = tris[f];
            const int i = t.group[0] == static_cast<int32_t>(gi) ? 0 : (t.group[1] == static_cast<int32_t>(gi) ? 1 : 2);
            const V3 n = m.normal(t.v[i]);
            const V3 os = projected(n, t.os), ot = projected(n, t.ot);
            members.clear();
            for (uint32_t other : g.faces) {
                const Tri& o = tris[other];
                const V3 os2 = projected(n, o.os), ot2 = projected(n, o.ot);
                const bool any = t.groupWithAny || o.groupWithAny;
                if (any || other == f || (dot(os, os2) > thresholdCos && dot(ot, ot2) > thresholdCos)) members.push_back(other);
            }
            std::sort(members.begin(), members.end());
            size_t which = 0;
            while (which < subGroups.size() && subGroups[which] != members) ++which;
            if (which == subGroups.size()) {
                subGroups.push_back(members);
                subFrames.push_back(evaluate(m, tris, members, g.vertex));
            }
            Frame out = subFrames[which];
            out.orient = g.orientPreserving;
            frames[static_cast<size_t>(t.source) * 3 + static_cast<size_t>(i)] = out;
        }
    }

    // ---- 6. degenerate triangles: the frame of the first good corner at the same vertex ----
    if (!degenerate.empty()) {
        std::unordered_map<uint32_t, size_t> firstGoodCorner;
        for (const Tri& t : tris) {
            for (int i = 0; i < 3; ++i) firstGoodCorner.try_emplace(t.v[i], static_cast<size_t>(t.source) * 3 + static_cast<size_t>(i));
        }
        for (uint32_t f : degenerate) {
            for (int i = 0; i < 3; ++i) {
                const auto found = firstGoodCorner.find(vertexOf[3 * f + static_cast<size_t>(i)]);
                if (found != firstGoodCorner.end()) frames[static_cast<size_t>(f) * 3 + static_cast<size_t>(i)] = frames[found->second];
            }
        }
    }
    for (size_t c = 0; c < corners; ++c) {
        tangents[4 * c + 0] = frames[c].os.x;
        tangents[4 * c + 1] = frames[c].os.y;
        tangents[4 * c + 2] = frames[c].os.z;
        tangents[4 * c + 3] = frames[c].orient ? 1.0f : -1.0f;
    }
    return true;
}

namespace {

// v / |v|, or (1, 0, 0) for a vector without length (TangentGen.mm:16-22)
float3 unitOrX(const float3& v) {
    const float len = length(v);
    if (len <= 1.0e-8f) return float3{1.0f, 0.0f, 0.0f};
    return float3{v.x / len, v.y / len, v.z / len};
}

// angle-weighted per-vertex tangents from the triangles' UV derivatives, mirrored and unmirrored sides kept apart and the heavier
// one taken (TangentGen.mm:24-119): what the reference falls back on when the library reports failure
void fallbackTangents(std::vector<SceneResources::MeshVertex>& vertices, const std::vector<uint32_t>& indices) {
    struct Sum {
        float3 tangent{0.0f, 0.0f, 0.0f}, bitangent{0.0f, 0.0f, 0.0f};
        float weight = 0.0f;
    };
    std::vector<Sum> positive(vertices.size()), negative(vertices.size());
    for (size_t i = 0; i + 2 < indices.size(); i += 3) {
        const uint32_t id[3] = {indices[i], indices[i + 1], indices[i + 2]};
        if (id[0] >= vertices.size() || id[1] >= vertices.size() || id[2] >= vertices.size()) continue;
        const SceneResources::MeshVertex* v[3] = {&vertices[id[0]], &vertices[id[1]], &vertices[id[2]]};
        const float3 edge1 = v[1]->position - v[0]->position, edge2 = v[2]->position - v[0]->position;
        const float du1 = v[1]->uv.x - v[0]->uv.x, dv1 = v[1]->uv.y - v[0]->uv.y, du2 = v[2]->uv.x - v[0]->uv.x, dv2 = v[2]->uv.y - v[0]->uv.y;
        const float denom = du1 * dv2 - dv1 * du2;
        if (std::fabs(denom) < 1.0e-8f) continue;
        const float r = 1.0f / denom;
        const float3 tangent = (edge1 * dv2 - edge2 * dv1) * r, bitangent = (edge2 * du1 - edge1 * du2) * r;
        for (int c = 0; c < 3; ++c) {
            const float3 a = unitOrX(v[(c + 1) % 3]->position - v[c]->position), b = unitOrX(v[(c + 2) % 3]->position - v[c]->position);
            const float angle = std::acos(std::min(std::max(dot(a, b), -1.0f), 1.0f));
            const float3 n = unitOrX(v[c]->normal);
            Sum& into = dot(cross(n, tangent), bitangent) < 0.0f ? negative[id[c]] : positive[id[c]];
            into.tangent = into.tangent + tangent * angle;
            into.bitangent = into.bitangent + bitangent * angle;
            into.weight += angle;
        }
    }
    for (size_t i = 0; i < vertices.size(); ++i) {
        const float3 n = unitOrX(vertices[i].normal);
        const Sum& best = positive[i].weight >= negative[i].weight ? positive[i] : negative[i];
        const float3 t = unitOrX(best.tangent - n * dot(n, best.tangent));
        const float w = dot(cross(n, t), best.bitangent) < 0.0f ? -1.0f : 1.0f;
        vertices[i].tangent = float4{t.x, t.y, t.z, w};
    }
}

}  // namespace

void GenerateTangents(std::vector<SceneResources::MeshVertex>& vertices, std::vector<uint32_t>& indices) {
    if (vertices.empty() || indices.empty() || indices.size() % 3 != 0) return;
    // one vertex per triangle corner (a mesh that already is one keeps its arrays)
    bool deindex = indices.size() != vertices.size();
    for (size_t i = 0; i < indices.size() && !deindex; ++i) deindex = indices[i] != i;
    if (deindex) {
        std::vector<SceneResources::MeshVertex> corners;
        corners.reserve(indices.size());
        for (uint32_t index : indices) {
            if (index >= vertices.size()) return;
            corners.push_back(vertices[index]);
        }
        vertices.swap(corners);
        for (size_t i = 0; i < indices.size(); ++i) indices[i] = static_cast<uint32_t>(i);
    }
    const size_t corners = vertices.size();
    std::vector<float> pos(corners * 3), nrm(corners * 3), uv(corners * 2), tan(corners * 4);
    for (size_t c = 0; c < corners; ++c) {
        const SceneResources::MeshVertex& v = vertices[c];
        const float3 n = unitOrX(v.normal);   // the library is handed unit normals (TangentGen.mm:145-155)
        pos[3 * c] = v.position.x, pos[3 * c + 1] = v.position.y, pos[3 * c + 2] = v.position.z;
        nrm[3 * c] = n.x, nrm[3 * c + 1] = n.y, nrm[3 * c + 2] = n.z;
        uv[2 * c] = v.uv.x, uv[2 * c + 1] = v.uv.y;
    }
    if (!GenerateTangentSpace(pos.data(), nrm.data(), uv.data(), corners / 3, tan.data())) {
        fallbackTangents(vertices, indices);
        return;
    }
    for (size_t c = 0; c < corners; ++c) vertices[c].tangent = float4{tan[4 * c], tan[4 * c + 1], tan[4 * c + 2], tan[4 * c + 3]};
}

}  // namespace ptr
