#include "mesh_loaders.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace ptr {
namespace {

// Flat-shading fallback used when all three corners lack a normal (SceneManager.mm:69-94).
void applyFallbackNormals(SceneResources::MeshVertex& v0, SceneResources::MeshVertex& v1,
                          SceneResources::MeshVertex& v2) {
    const bool hasNormal = length(v0.normal) > 0.0f || length(v1.normal) > 0.0f || length(v2.normal) > 0.0f;
    if (hasNormal) {
        return;
    }
    const float3 n = cross(v1.position - v0.position, v2.position - v0.position);
    if (!(length(n) > 0.0f)) {
        return;
    }
    const float3 unit = normalize(n);
    v0.normal = v1.normal = v2.normal = unit;
}

struct CornerKey {
    int p, n, t;
    bool operator==(const CornerKey& o) const { return p == o.p && n == o.n && t == o.t; }
};
struct CornerHash {
    size_t operator()(const CornerKey& k) const {
        uint64_t h = static_cast<uint32_t>(k.p) * 0x9E3779B97F4A7C15ull;
        h ^= (static_cast<uint64_t>(static_cast<uint32_t>(k.n)) + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= (static_cast<uint64_t>(static_cast<uint32_t>(k.t)) + 0x165667B1ull) * 0x27D4EB2F165667C5ull;
        return static_cast<size_t>(h ^ (h >> 29));
    }
};

// Parse one "p", "p/t", "p//n" or "p/t/n" face corner; OBJ indices are 1-based, negative = relative.
bool parseCorner(const char*& s, int counts[3], CornerKey& key) {
    int vals[3] = {0, 0, 0};
    bool have[3] = {false, false, false};
    for (int field = 0; field < 3; ++field) {
        char* end = nullptr;
        const long v = std::strtol(s, &end, 10);
        if (end != s) {
            vals[field] = static_cast<int>(v);
            have[field] = true;
            s = end;
        }
        if (*s == '/') {
            ++s;
        } else {
            break;
        }
    }
    if (!have[0]) return false;
    auto resolve = [](int v, int count) { return v > 0 ? v - 1 : (v < 0 ? count + v : -1); };
    key.p = resolve(vals[0], counts[0]);
    key.t = have[1] ? resolve(vals[1], counts[1]) : -1;
    key.n = have[2] ? resolve(vals[2], counts[2]) : -1;
    return true;
}

}  // namespace

bool LoadObjMesh(const std::string& path, LoadedMeshData& out, std::string& error) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        error = "Failed to parse OBJ file: " + path;
        return false;
    }
    std::vector<float> pos, nrm, tex;
    std::vector<CornerKey> corners;  // triangulated, 3 per face
    std::vector<char> line(1 << 16);
    std::vector<CornerKey> poly;
    while (fgets(line.data(), static_cast<int>(line.size()), f)) {
        const char* s = line.data();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* e = nullptr;
            const char* p = s + 2;
            for (int i = 0; i < 3; ++i) {
                pos.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (s[0] == 'v' && s[1] == 'n') {
            char* e = nullptr;
            const char* p = s + 3;
            for (int i = 0; i < 3; ++i) {
                nrm.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (s[0] == 'v' && s[1] == 't') {
            char* e = nullptr;
            const char* p = s + 3;
            for (int i = 0; i < 2; ++i) {
                tex.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            int counts[3] = {static_cast<int>(pos.size() / 3), static_cast<int>(tex.size() / 2),
                             static_cast<int>(nrm.size() / 3)};
            poly.clear();
            const char* p = s + 2;
            while (true) {
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                CornerKey key{};
                if (!parseCorner(p, counts, key)) break;
                poly.push_back(key);
            }
            // Triangulation as the reference's OBJ library does it (tinyobjloader, external/tinyobjloader/tiny_obj_loader.h:1520-1620,
            // called with triangulate = true, SceneManager.mm:100): a quad is cut along its shorter diagonal - (0 1 2)(0 2 3) when
            // |v2 - v0|^2 < |v3 - v1|^2, else (0 1 3)(1 2 3); larger polygons are fanned from their first corner (the library clips
            // ears there: the same triangles for the convex polygons exporters write).
            bool cut13 = false;
            if (poly.size() == 4) {
                const float* v[4];
                bool known = true;
                for (int k = 0; k < 4; ++k) {
                    known = known && poly[static_cast<size_t>(k)].p >= 0 && static_cast<size_t>(poly[static_cast<size_t>(k)].p) * 3 + 2 < pos.size();
                    v[k] = known ? &pos[static_cast<size_t>(poly[static_cast<size_t>(k)].p) * 3] : nullptr;
                }
                if (known) {
                    const float ax = v[2][0] - v[0][0], ay = v[2][1] - v[0][1], az = v[2][2] - v[0][2];
                    const float bx = v[3][0] - v[1][0], by = v[3][1] - v[1][1], bz = v[3][2] - v[1][2];
                    cut13 = !(ax * ax + ay * ay + az * az < bx * bx + by * by + bz * bz);
                }
            }
            if (cut13) {
                for (int k : {0, 1, 3, 1, 2, 3}) corners.push_back(poly[static_cast<size_t>(k)]);
            } else {
                for (size_t k = 1; k + 1 < poly.size(); ++k) {
                    corners.push_back(poly[0]);
                    corners.push_back(poly[k]);
                    corners.push_back(poly[k + 1]);
                }
            }
        }
    }
    fclose(f);

    if (pos.empty()) {
        error = "OBJ file contains no vertex positions: " + path;
        return false;
    }
    if (corners.empty()) {
        error = "OBJ file contains no triangle data: " + path;
        return false;
    }

    out.vertices.clear();
    out.indices.clear();
    out.indices.reserve(corners.size());
    std::unordered_map<CornerKey, uint32_t, CornerHash> lookup;
    lookup.reserve(corners.size());
    const int posCount = static_cast<int>(pos.size() / 3);
    for (const CornerKey& key : corners) {
        if (key.p < 0 || key.p >= posCount) {
            error = "OBJ references a position index that is out of range";
            return false;
        }
        auto it = lookup.find(key);
        if (it == lookup.end()) {
            SceneResources::MeshVertex v{};  // normal defaults to (0,1,0): reference quirk Q3
            v.position = {pos[key.p * 3 + 0], pos[key.p * 3 + 1], pos[key.p * 3 + 2]};
            if (key.n >= 0 && static_cast<size_t>(key.n) * 3 + 2 < nrm.size()) {
                v.normal = {nrm[key.n * 3 + 0], nrm[key.n * 3 + 1], nrm[key.n * 3 + 2]};
            }
            if (key.t >= 0 && static_cast<size_t>(key.t) * 2 + 1 < tex.size()) {
                v.uv.x = tex[key.t * 2 + 0];
                v.uv.y = tex[key.t * 2 + 1];
            }
            const uint32_t idx = static_cast<uint32_t>(out.vertices.size());
            out.vertices.push_back(v);
            it = lookup.emplace(key, idx).first;
        }
        out.indices.push_back(it->second);
    }
    for (size_t i = 0; i + 2 < out.indices.size(); i += 3) {
        applyFallbackNormals(out.vertices[out.indices[i]], out.vertices[out.indices[i + 1]],
                             out.vertices[out.indices[i + 2]]);
    }
    return true;
}

namespace {

enum class PlyType { Invalid, I8, U8, I16, U16, I32, U32, F32, F64 };

PlyType plyTypeFromString(const std::string& s) {
    if (s == "char" || s == "int8") return PlyType::I8;
    if (s == "uchar" || s == "uint8") return PlyType::U8;
    if (s == "short" || s == "int16") return PlyType::I16;
    if (s == "ushort" || s == "uint16") return PlyType::U16;
    if (s == "int" || s == "int32") return PlyType::I32;
    if (s == "uint" || s == "uint32") return PlyType::U32;
    if (s == "float" || s == "float32") return PlyType::F32;
    if (s == "double" || s == "float64") return PlyType::F64;
    return PlyType::Invalid;
}

size_t plyTypeSize(PlyType t) {
    switch (t) {
        case PlyType::I8: case PlyType::U8: return 1;
        case PlyType::I16: case PlyType::U16: return 2;
        case PlyType::I32: case PlyType::U32: case PlyType::F32: return 4;
        case PlyType::F64: return 8;
        default: return 0;
    }
}

struct PlyProperty {
    std::string name;
    PlyType type = PlyType::Invalid;
    bool isList = false;
    PlyType listCountType = PlyType::Invalid;
};

struct PlyElement {
    std::string name;
    size_t count = 0;
    std::vector<PlyProperty> props;
};

double readBinary(std::istream& in, PlyType t) {
    char buf[8] = {0};
    in.read(buf, static_cast<std::streamsize>(plyTypeSize(t)));
    switch (t) {
        case PlyType::I8: { int8_t v; std::memcpy(&v, buf, 1); return v; }
        case PlyType::U8: { uint8_t v; std::memcpy(&v, buf, 1); return v; }
        case PlyType::I16: { int16_t v; std::memcpy(&v, buf, 2); return v; }
        case PlyType::U16: { uint16_t v; std::memcpy(&v, buf, 2); return v; }
        case PlyType::I32: { int32_t v; std::memcpy(&v, buf, 4); return v; }
        case PlyType::U32: { uint32_t v; std::memcpy(&v, buf, 4); return v; }
        case PlyType::F32: { float v; std::memcpy(&v, buf, 4); return v; }
        case PlyType::F64: { double v; std::memcpy(&v, buf, 8); return v; }
        default: return 0.0;
    }
}

}  // namespace

bool LoadPlyMesh(const std::string& path, LoadedMeshData& out, std::string& error) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) {
        error = "Failed to open PLY file: " + path;
        return false;
    }
    std::string line;
    if (!std::getline(in, line) || line.rfind("ply", 0) != 0) {
        error = "PLY header parsing failed: " + path;
        return false;
    }
    bool ascii = false, binaryLE = false;
    std::vector<PlyElement> elements;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string word;
        ls >> word;
        if (word == "format") {
            std::string fmt;
            ls >> fmt;
            ascii = (fmt == "ascii");
            binaryLE = (fmt == "binary_little_endian");
        } else if (word == "element") {
            PlyElement e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (word == "property" && !elements.empty()) {
            PlyProperty p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.isList = true;
                p.listCountType = plyTypeFromString(ct);
                p.type = plyTypeFromString(it);
            } else {
                p.type = plyTypeFromString(t);
                ls >> p.name;
            }
            elements.back().props.push_back(p);
        } else if (word == "end_header") {
            break;
        }
    }
    if (!ascii && !binaryLE) {
        error = "PLY header parsing failed: " + path + " (only ascii / binary_little_endian supported)";
        return false;
    }

    out.vertices.clear();
    out.indices.clear();
    bool sawPositions = false, sawFaces = false;
    size_t vertsPerFace = 0;

    auto readValue = [&](PlyType t) -> double {
        if (ascii) {
            double v = 0.0;
            in >> v;
            return v;
        }
        return readBinary(in, t);
    };

    for (const PlyElement& el : elements) {
        if (el.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t i = 0; i < el.props.size(); ++i) {
                const std::string& n = el.props[i].name;
                const int idx = static_cast<int>(i);
                if (n == "x") ix = idx; else if (n == "y") iy = idx; else if (n == "z") iz = idx;
                else if (n == "nx") inx = idx; else if (n == "ny") iny = idx; else if (n == "nz") inz = idx;
                else if ((n == "u" || n == "s" || n == "texture_u") && iu < 0) iu = idx;
                else if ((n == "v" || n == "t" || n == "texture_v") && iv < 0) iv = idx;
            }
            if (ix < 0 || iy < 0 || iz < 0) {
                error = "PLY file is missing vertex position data";
                return false;
            }
            for (int i : {ix, iy, iz}) {
                if (el.props[i].type != PlyType::F32 && el.props[i].type != PlyType::F64) {
                    error = "PLY vertex positions must be stored as float or double";
                    return false;
                }
            }
            sawPositions = true;
            const bool hasNormals = inx >= 0 && iny >= 0 && inz >= 0;
            out.vertices.resize(el.count);
            std::vector<double> vals(el.props.size());
            for (size_t v = 0; v < el.count; ++v) {
                for (size_t p = 0; p < el.props.size(); ++p) {
                    if (el.props[p].isList) {
                        const size_t n = static_cast<size_t>(readValue(el.props[p].listCountType));
                        for (size_t k = 0; k < n; ++k) readValue(el.props[p].type);
                        vals[p] = 0.0;
                    } else {
                        vals[p] = readValue(el.props[p].type);
                    }
                }
                SceneResources::MeshVertex mv{};
                mv.position = {static_cast<float>(vals[ix]), static_cast<float>(vals[iy]), static_cast<float>(vals[iz])};
                mv.normal = {0.0f, 0.0f, 0.0f};  // PLY zeroes normals so the flat fallback can fire
                if (hasNormals) {
                    mv.normal = {static_cast<float>(vals[inx]), static_cast<float>(vals[iny]), static_cast<float>(vals[inz])};
                }
                if (iu >= 0 && iv >= 0) {
                    mv.uv.x = static_cast<float>(vals[iu]);
                    mv.uv.y = static_cast<float>(vals[iv]);
                }
                out.vertices[v] = mv;
            }
        } else if (el.name == "face") {
            int il = -1;
            for (size_t i = 0; i < el.props.size(); ++i) {
                if (el.props[i].isList && (el.props[i].name == "vertex_indices" || el.props[i].name == "vertex_index")) {
                    il = static_cast<int>(i);
                }
            }
            if (il < 0) {
                error = "PLY file is missing face index data";
                return false;
            }
            if (el.count == 0) {
                error = "PLY contains no faces";
                return false;
            }
            sawFaces = true;
            std::vector<uint32_t> face;
            for (size_t fidx = 0; fidx < el.count; ++fidx) {
                for (size_t p = 0; p < el.props.size(); ++p) {
                    if (el.props[p].isList) {
                        const size_t n = static_cast<size_t>(readValue(el.props[p].listCountType));
                        if (static_cast<int>(p) == il) {
                            face.resize(n);
                            for (size_t k = 0; k < n; ++k) {
                                const double v = readValue(el.props[p].type);
                                if (v < 0.0) {
                                    error = "PLY face index is negative";
                                    return false;
                                }
                                face[k] = static_cast<uint32_t>(v);
                            }
                        } else {
                            for (size_t k = 0; k < n; ++k) readValue(el.props[p].type);
                        }
                    } else {
                        readValue(el.props[p].type);
                    }
                }
                if (vertsPerFace == 0) vertsPerFace = face.size();
                if (face.size() != vertsPerFace) {
                    error = "PLY uses variable-length face lists which are not currently supported";
                    return false;
                }
                if (face.size() < 3) {
                    error = "PLY face has fewer than three indices";
                    return false;
                }
                for (uint32_t idx : face) {
                    if (idx >= out.vertices.size()) {
                        error = "PLY face index is out of range";
                        return false;
                    }
                }
                for (size_t k = 1; k + 1 < face.size(); ++k) {
                    out.indices.push_back(face[0]);
                    out.indices.push_back(face[k]);
                    out.indices.push_back(face[k + 1]);
                }
            }
        } else {
            // skip unknown elements
            for (size_t i = 0; i < el.count; ++i) {
                for (const PlyProperty& p : el.props) {
                    if (p.isList) {
                        const size_t n = static_cast<size_t>(readValue(p.listCountType));
                        for (size_t k = 0; k < n; ++k) readValue(p.type);
                    } else {
                        readValue(p.type);
                    }
                }
            }
        }
        if (!in) {
            error = "PLY payload read error: unexpected end of file";
            return false;
        }
    }
    if (!sawPositions) {
        error = "PLY file is missing vertex position data";
        return false;
    }
    if (out.vertices.empty()) {
        error = "PLY contains no vertices";
        return false;
    }
    if (!sawFaces) {
        error = "PLY file is missing face index data";
        return false;
    }
    for (size_t i = 0; i + 2 < out.indices.size(); i += 3) {
        applyFallbackNormals(out.vertices[out.indices[i]], out.vertices[out.indices[i + 1]],
                             out.vertices[out.indices[i + 2]]);
    }
    return true;
}

}  // namespace ptr
