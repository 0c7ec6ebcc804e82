// glTF 2.0 / GLB import: geometry, node hierarchy, embedded camera and factor-level PBR materials.
//
// Behaviour follows the reference loader (src/assets/GltfLoader.mm): what is read from the document
// (792-1237), how a material becomes a 576 B MaterialData of type PbrMetallicRoughness (650-788), how a
// triangle primitive becomes a SceneResources mesh incl. area-weighted normals when NORMAL is absent
// (1239-1466) and the depth-first node walk with column-major TRS composition (1468-1534).  Image decoding is
// (textures: decoded by csrc/host/image_decoders.cpp and handed to SceneResources::addTexture; they are sampled only by the
// Metal-semantics PBR model, PtrSettings.metalSemantics bit PTR_METAL_PBR - see below)
// not done here: the parity oracle (the Embree backend) never samples material textures, so every texture slot
// is recorded as "invalid" exactly as the reference leaves it when no Metal device exists
// (SceneResources.mm:1279-1284); texture coordinate sets and KHR_texture_transform rows are still stored.
// The JSON reader below replaces NSJSONSerialization.
#include "gltf_loader.h"
#include "tangent_space.h"

#include "image_decoders.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>

namespace fs = std::filesystem;

namespace ptr {
namespace {

constexpr uint32_t kNoTexture = 0xFFFFFFFFu;
constexpr uint32_t kFlagDisableOrm = 1u;   // MetalShaderTypes.h kMaterialFlagDisableOrm

// ------------------------------------------------------------------------------------------ JSON
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean = false;
    double number = 0.0;
    std::string text;
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> members;

    const Json* get(const char* key) const {
        if (kind != Object) return nullptr;
        for (const auto& m : members) {
            if (m.first == key) return &m.second;
        }
        return nullptr;
    }
    const Json* object(const char* key) const {
        const Json* v = get(key);
        return (v && v->kind == Object) ? v : nullptr;
    }
    const Json* array(const char* key) const {
        const Json* v = get(key);
        return (v && v->kind == Array) ? v : nullptr;
    }
    // NSNumber semantics of the reference getters: numbers and booleans both count as numbers
    bool numeric(const char* key, double& out) const {
        const Json* v = get(key);
        if (!v) return false;
        if (v->kind == Number) {
            out = v->number;
            return true;
        }
        if (v->kind == Bool) {
            out = v->boolean ? 1.0 : 0.0;
            return true;
        }
        return false;
    }
    int intOr(const char* key, int fallback) const {
        double d;
        return numeric(key, d) ? static_cast<int>(d) : fallback;
    }
    float floatOr(const char* key, float fallback) const {
        double d;
        return numeric(key, d) ? static_cast<float>(d) : fallback;
    }
    bool boolOr(const char* key, bool fallback) const {
        double d;
        return numeric(key, d) ? d != 0.0 : fallback;
    }
    std::string stringOr(const char* key) const {
        const Json* v = get(key);
        return (v && v->kind == String) ? v->text : std::string();
    }
    bool floats(const char* key, std::vector<float>& out) const {
        const Json* v = array(key);
        if (!v) return false;
        out.resize(v->items.size());
        for (size_t i = 0; i < v->items.size(); ++i) {
            const Json& e = v->items[i];
            out[i] = e.kind == Number ? static_cast<float>(e.number) : (e.kind == Bool && e.boolean ? 1.0f : 0.0f);
        }
        return true;
    }
};

class JsonReader {
public:
    JsonReader(const char* begin, const char* end) : m_p(begin), m_end(end) {}
    bool parse(Json& out) {
        if (!value(out, 0)) return false;
        skipSpace();
        return m_p == m_end || *m_p == '\0';
    }

private:
    const char* m_p;
    const char* m_end;
    static constexpr int kMaxDepth = 128;

    void skipSpace() {
        while (m_p < m_end && (*m_p == ' ' || *m_p == '\t' || *m_p == '\n' || *m_p == '\r')) ++m_p;
    }
    bool literal(const char* word) {
        const size_t n = std::strlen(word);
        if (static_cast<size_t>(m_end - m_p) < n || std::strncmp(m_p, word, n) != 0) return false;
        m_p += n;
        return true;
    }
    static void appendUtf8(std::string& s, uint32_t cp) {
        if (cp < 0x80) {
            s.push_back(static_cast<char>(cp));
        } else if (cp < 0x800) {
            s.push_back(static_cast<char>(0xC0 | (cp >> 6)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s.push_back(static_cast<char>(0xE0 | (cp >> 12)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else {
            s.push_back(static_cast<char>(0xF0 | (cp >> 18)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t& out) {
        if (m_end - m_p < 4) return false;
        out = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *m_p++;
            out <<= 4;
            if (c >= '0' && c <= '9') out |= static_cast<uint32_t>(c - '0');
            else if (c >= 'a' && c <= 'f') out |= static_cast<uint32_t>(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') out |= static_cast<uint32_t>(c - 'A' + 10);
            else return false;
        }
        return true;
    }
    bool string(std::string& out) {
        if (m_p >= m_end || *m_p != '"') return false;
        ++m_p;
        out.clear();
        while (m_p < m_end) {
            const char c = *m_p++;
            if (c == '"') return true;
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (m_p >= m_end) return false;
            const char e = *m_p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF && m_end - m_p >= 6 && m_p[0] == '\\' && m_p[1] == 'u') {
                        m_p += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    appendUtf8(out, cp);
                    break;
                }
                default: return false;
            }
        }
        return false;
    }
    bool value(Json& out, int depth) {
        if (depth > kMaxDepth) return false;
        skipSpace();
        if (m_p >= m_end) return false;
        const char c = *m_p;
        if (c == '{') {
            ++m_p;
            out.kind = Json::Object;
            skipSpace();
            if (m_p < m_end && *m_p == '}') {
                ++m_p;
                return true;
            }
            while (true) {
                skipSpace();
                std::string key;
                if (!string(key)) return false;
                skipSpace();
                if (m_p >= m_end || *m_p != ':') return false;
                ++m_p;
                out.members.emplace_back(std::move(key), Json{});
                if (!value(out.members.back().second, depth + 1)) return false;
                skipSpace();
                if (m_p >= m_end) return false;
                if (*m_p == ',') {
                    ++m_p;
                    continue;
                }
                if (*m_p == '}') {
                    ++m_p;
                    return true;
                }
                return false;
            }
        }
        if (c == '[') {
            ++m_p;
            out.kind = Json::Array;
            skipSpace();
            if (m_p < m_end && *m_p == ']') {
                ++m_p;
                return true;
            }
            while (true) {
                out.items.emplace_back();
                if (!value(out.items.back(), depth + 1)) return false;
                skipSpace();
                if (m_p >= m_end) return false;
                if (*m_p == ',') {
                    ++m_p;
                    continue;
                }
                if (*m_p == ']') {
                    ++m_p;
                    return true;
                }
                return false;
            }
        }
        if (c == '"') {
            out.kind = Json::String;
            return string(out.text);
        }
        if (c == 't') {
            out.kind = Json::Bool;
            out.boolean = true;
            return literal("true");
        }
        if (c == 'f') {
            out.kind = Json::Bool;
            out.boolean = false;
            return literal("false");
        }
        if (c == 'n') {
            out.kind = Json::Null;
            return literal("null");
        }
        // number: copy the JSON number token and let strtod do the conversion
        const char* start = m_p;
        if (m_p < m_end && *m_p == '-') ++m_p;
        while (m_p < m_end && (std::isdigit(static_cast<unsigned char>(*m_p)) || *m_p == '.' || *m_p == 'e' || *m_p == 'E' ||
                               *m_p == '+' || *m_p == '-')) {
            ++m_p;
        }
        if (m_p == start) return false;
        const std::string token(start, m_p);
        char* endp = nullptr;
        out.number = std::strtod(token.c_str(), &endp);
        if (endp == token.c_str() || *endp != '\0') return false;
        out.kind = Json::Number;
        return true;
    }
};

// ------------------------------------------------------------------------------------------ bytes
bool readFileBytes(const fs::path& path, std::vector<uint8_t>& out, std::string& error) {
    std::error_code ec;
    if (!fs::exists(path, ec)) {
        error = "glTF file not found: " + path.string();
        return false;
    }
    std::ifstream stream(path, std::ios::binary);
    if (!stream.is_open()) {
        error = "Failed to open glTF file: " + path.string();
        return false;
    }
    stream.seekg(0, std::ios::end);
    const std::streamsize size = stream.tellg();
    if (size <= 0) {
        error = "glTF file is empty: " + path.string();
        return false;
    }
    stream.seekg(0, std::ios::beg);
    out.resize(static_cast<size_t>(size));
    stream.read(reinterpret_cast<char*>(out.data()), size);
    if (!stream) {
        error = "Failed to read glTF file: " + path.string();
        return false;
    }
    return true;
}

// data:[<mime>][;base64],<payload> — only base64 payloads are accepted (like the reference)
bool decodeDataUri(const std::string& uri, std::vector<uint8_t>& out) {
    if (uri.rfind("data:", 0) != 0) return false;
    const size_t comma = uri.find(',');
    if (comma == std::string::npos) return false;
    if (uri.substr(5, comma - 5).find(";base64") == std::string::npos) return false;
    auto sextet = [](unsigned char c) -> int {
        if (c >= 'A' && c <= 'Z') return c - 'A';
        if (c >= 'a' && c <= 'z') return c - 'a' + 26;
        if (c >= '0' && c <= '9') return c - '0' + 52;
        if (c == '+' || c == '-') return 62;
        if (c == '/' || c == '_') return 63;
        return -1;
    };
    out.clear();
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = comma + 1; i < uri.size(); ++i) {
        const unsigned char c = static_cast<unsigned char>(uri[i]);
        if (c == '=') break;
        if (std::isspace(c)) continue;
        const int v = sextet(c);
        if (v < 0) return false;
        acc = (acc << 6) | static_cast<uint32_t>(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back(static_cast<uint8_t>((acc >> bits) & 0xFFu));
        }
    }
    return true;
}

std::string lowerAscii(std::string s) {
    for (char& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

bool containsNoCase(const std::string& text, const std::string& needle) {
    return !needle.empty() && lowerAscii(text).find(lowerAscii(needle)) != std::string::npos;
}

// ------------------------------------------------------------------------------------------ document model
struct BufferView {
    int buffer = -1;
    size_t offset = 0, length = 0, stride = 0;
};

struct Accessor {
    int view = -1;
    size_t offset = 0, count = 0;
    int componentType = 0;
    std::string type;
    bool normalized = false;
};

struct TextureBinding {
    int index = -1;
    int texCoord = 0;
    float offset[2] = {0.0f, 0.0f};
    float scale[2] = {1.0f, 1.0f};
    float rotation = 0.0f;
};

struct SourceMaterial {
    std::string name;
    float baseColor[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float metallic = 1.0f, roughness = 1.0f;
    std::string alphaMode = "OPAQUE";
    float alphaCutoff = 0.5f;
    bool doubleSided = false;
    TextureBinding baseColorTex, metalRoughTex, normalTex, occlusionTex, emissiveTex, transmissionTex;
    float normalScale = 1.0f, occlusionStrength = 1.0f;
    float emissive[3] = {0.0f, 0.0f, 0.0f};
    float emissiveStrength = 1.0f;
    float transmission = 0.0f;
    bool hasVolume = false;
    float thickness = 0.0f;
    float attenuationColor[3] = {1.0f, 1.0f, 1.0f};
    float attenuationDistance = std::numeric_limits<float>::infinity();
    bool hasIor = false;
    float ior = 1.5f;
    bool disableOrm = false;
};

struct Primitive {
    int material = -1, position = -1, normal = -1, texcoord = -1, texcoord1 = -1, tangent = -1, indices = -1, mode = 4;
};

struct MeshDef {
    std::string name;
    std::vector<Primitive> primitives;
};

struct CameraDef {
    bool perspective = false;
    float yfov = 0.0f;
};

struct NodeDef {
    std::string name;
    int mesh = -1, camera = -1;
    std::vector<int> children;
    float translation[3] = {0.0f, 0.0f, 0.0f};
    float rotation[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    float scale[3] = {1.0f, 1.0f, 1.0f};
    bool hasMatrix = false;
    float4x4 matrix = float4x4::identity();
};

int clampTexCoordSet(int set) { return set < 0 ? 0 : std::min(set, 1); }

void parseBinding(const Json* info, TextureBinding& out) {
    if (!info || info->kind != Json::Object) return;
    out.index = info->intOr("index", -1);
    out.texCoord = clampTexCoordSet(info->intOr("texCoord", 0));
    const Json* ext = info->object("extensions");
    const Json* xf = ext ? ext->object("KHR_texture_transform") : nullptr;
    if (!xf) return;
    std::vector<float> v;
    if (xf->floats("offset", v) && v.size() >= 2) {
        out.offset[0] = v[0];
        out.offset[1] = v[1];
    }
    if (xf->floats("scale", v) && v.size() >= 2) {
        out.scale[0] = v[0];
        out.scale[1] = v[1];
    }
    out.rotation = xf->floatOr("rotation", 0.0f);
    out.texCoord = clampTexCoordSet(xf->intOr("texCoord", out.texCoord));
}

size_t componentsOf(const std::string& type) {
    if (type == "SCALAR") return 1;
    if (type == "VEC2") return 2;
    if (type == "VEC3") return 3;
    if (type == "VEC4") return 4;
    return 0;
}

size_t componentSize(int componentType) {
    switch (componentType) {
        case 5126: case 5125: return 4;
        case 5123: case 5122: return 2;
        case 5121: case 5120: return 1;
        default: return 0;
    }
}

struct Document {
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<BufferView> views;
    std::vector<Accessor> accessors;

    // start of the accessor's first element + element stride; false when anything is out of range
    bool locate(const Accessor& a, const uint8_t*& base, size_t& stride, size_t& available) const {
        if (a.view < 0 || a.view >= static_cast<int>(views.size())) return false;
        const BufferView& v = views[a.view];
        if (v.buffer < 0 || v.buffer >= static_cast<int>(buffers.size())) return false;
        const std::vector<uint8_t>& data = buffers[v.buffer];
        const size_t offset = v.offset + a.offset;
        if (offset >= data.size()) return false;
        const size_t cs = componentSize(a.componentType), n = componentsOf(a.type);
        if (cs == 0 || n == 0) return false;
        base = data.data() + offset;
        stride = v.stride != 0 ? v.stride : cs * n;
        available = data.size() - offset;
        return true;
    }

    static float component(const uint8_t* p, int componentType, bool normalized) {
        switch (componentType) {
            case 5126: { float v; std::memcpy(&v, p, 4); return v; }
            case 5125: { uint32_t v; std::memcpy(&v, p, 4); return normalized ? static_cast<float>(v) / 4294967295.0f : static_cast<float>(v); }
            case 5123: { uint16_t v; std::memcpy(&v, p, 2); return normalized ? static_cast<float>(v) / 65535.0f : static_cast<float>(v); }
            case 5121: { return normalized ? static_cast<float>(*p) / 255.0f : static_cast<float>(*p); }
            case 5122: { int16_t v; std::memcpy(&v, p, 2); return normalized ? std::max(-1.0f, static_cast<float>(v) / 32767.0f) : static_cast<float>(v); }
            case 5120: { int8_t v; std::memcpy(&v, p, 1); return normalized ? std::max(-1.0f, static_cast<float>(v) / 127.0f) : static_cast<float>(v); }
            default: return 0.0f;
        }
    }

    bool readFloats(const Accessor& a, size_t components, std::vector<float>& out) const {
        if (componentsOf(a.type) != components) return false;
        const uint8_t* base = nullptr;
        size_t stride = 0, available = 0;
        if (!locate(a, base, stride, available)) return false;
        const size_t cs = componentSize(a.componentType);
        if (a.count > 0 && (a.count - 1) * stride + cs * components > available) return false;   // truncated buffer
        out.resize(a.count * components);
        for (size_t i = 0; i < a.count; ++i) {
            for (size_t c = 0; c < components; ++c) out[i * components + c] = component(base + i * stride + c * cs, a.componentType, a.normalized);
        }
        return true;
    }

    bool readIndices(const Accessor& a, std::vector<uint32_t>& out) const {
        const uint8_t* base = nullptr;
        size_t stride = 0, available = 0;
        if (!locate(a, base, stride, available)) return false;
        const size_t cs = componentSize(a.componentType);
        if (a.componentType != 5125 && a.componentType != 5123 && a.componentType != 5121) return false;
        if (a.count > 0 && (a.count - 1) * stride + cs > available) return false;
        out.resize(a.count);
        for (size_t i = 0; i < a.count; ++i) {
            const uint8_t* p = base + i * stride;
            if (a.componentType == 5125) {
                std::memcpy(&out[i], p, 4);
            } else if (a.componentType == 5123) {
                uint16_t v;
                std::memcpy(&v, p, 2);
                out[i] = v;
            } else {
                out[i] = *p;
            }
        }
        return true;
    }
};

// ------------------------------------------------------------------------------------------ transforms
float4x4 translationMatrix(const float* t) {
    float4x4 m = float4x4::identity();
    m.columns[3] = {t[0], t[1], t[2], 1.0f};
    return m;
}

float4x4 scaleMatrix(const float* s) {
    float4x4 m = float4x4::identity();
    m.columns[0].x = s[0];
    m.columns[1].y = s[1];
    m.columns[2].z = s[2];
    return m;
}

float4x4 rotationMatrix(const float* q) {   // unit quaternion (x, y, z, w), column-major result
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    float4x4 m;
    m.columns[0] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy), 0.0f};
    m.columns[1] = {2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx), 0.0f};
    m.columns[2] = {2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy), 0.0f};
    m.columns[3] = {0.0f, 0.0f, 0.0f, 1.0f};
    return m;
}

float4x4 localMatrix(const NodeDef& n) {
    if (n.hasMatrix) return n.matrix;
    return mul(translationMatrix(n.translation), mul(rotationMatrix(n.rotation), scaleMatrix(n.scale)));
}

float3 transformPoint(const float4x4& m, const float3& p) {
    const float4 r = mul(m, float4(p, 1.0f));
    return {r.x, r.y, r.z};
}

float3 transformDirection(const float4x4& m, const float3& d) {
    const float4 r = mul(m, float4(d, 0.0f));
    return {r.x, r.y, r.z};
}

// ------------------------------------------------------------------------------------------ materials
void transformRows(const TextureBinding& b, float* row0, float* row1) {
    const float c = std::cos(b.rotation), s = std::sin(b.rotation);
    row0[0] = c * b.scale[0], row0[1] = -s * b.scale[1], row0[2] = b.offset[0], row0[3] = 0.0f;
    row1[0] = s * b.scale[0], row1[1] = c * b.scale[1], row1[2] = b.offset[1], row1[3] = 0.0f;
}

float clamp01(float v) { return std::min(std::max(v, 0.0f), 1.0f); }

PtrMaterial buildMaterial(const SourceMaterial& src, const GltfLoadOptions& options) {
    PtrMaterial m;
    std::memset(&m, 0, sizeof(m));
    const float roughness = clamp01(src.roughness);
    const float ior = std::min(std::max(src.hasIor ? src.ior : 1.5f, 1.0f), 3.0f);
    m.baseColorRoughness[0] = src.baseColor[0];
    m.baseColorRoughness[1] = src.baseColor[1];
    m.baseColorRoughness[2] = src.baseColor[2];
    m.baseColorRoughness[3] = roughness;
    m.typeEta[0] = static_cast<float>(PTR_MAT_PBR);
    m.typeEta[1] = ior;
    m.typeEta[2] = src.doubleSided ? 1.0f : 0.0f;
    m.typeEta[3] = src.hasVolume ? std::max(src.thickness, 0.0f) : 0.0f;
    const float emissiveScale = std::max(src.emissiveStrength * options.emissiveScale, 0.0f);
    for (int c = 0; c < 3; ++c) m.emission[c] = src.emissive[c] * emissiveScale;
    if (src.hasVolume && std::isfinite(src.attenuationDistance) && src.attenuationDistance > 0.0f) {
        for (int c = 0; c < 3; ++c) {
            const float colour = std::min(std::max(src.attenuationColor[c], 1.0e-6f), 1.0f);
            m.dielectricSigmaA[c] = std::max(-std::log(colour) / src.attenuationDistance, 0.0f);
        }
    }
    m.carpaintBaseTint[0] = m.carpaintBaseTint[1] = m.carpaintBaseTint[2] = 1.0f;
    for (int i = 0; i < 4; ++i) {
        m.textureIndices0[i] = kNoTexture;   // base colour, metallic-roughness, normal, occlusion
        m.textureIndices1[i] = kNoTexture;   // emissive, transmission, unused, unused
    }
    m.textureUvSet0[0] = static_cast<uint32_t>(clampTexCoordSet(src.baseColorTex.texCoord));
    m.textureUvSet0[1] = static_cast<uint32_t>(clampTexCoordSet(src.metalRoughTex.texCoord));
    m.textureUvSet0[2] = static_cast<uint32_t>(clampTexCoordSet(src.normalTex.texCoord));
    m.textureUvSet0[3] = static_cast<uint32_t>(clampTexCoordSet(src.occlusionTex.texCoord));
    m.textureUvSet1[0] = static_cast<uint32_t>(clampTexCoordSet(src.emissiveTex.texCoord));
    m.textureUvSet1[1] = static_cast<uint32_t>(clampTexCoordSet(src.transmissionTex.texCoord));
    const TextureBinding* order[6] = {&src.baseColorTex, &src.metalRoughTex, &src.normalTex,
                                      &src.occlusionTex, &src.emissiveTex,   &src.transmissionTex};
    for (int i = 0; i < 6; ++i) transformRows(*order[i], m.textureTransform[2 * i], m.textureTransform[2 * i + 1]);
    m.pbrParams[0] = clamp01(src.metallic);
    m.pbrParams[1] = roughness;
    m.pbrParams[2] = clamp01(src.occlusionStrength);
    m.pbrParams[3] = std::max(src.normalScale, 0.0f);
    m.materialFlags = src.disableOrm ? kFlagDisableOrm : 0u;
    int alphaMode = 0;
    if (src.alphaMode == "MASK") alphaMode = 1;
    else if (src.alphaMode == "BLEND") alphaMode = 2;
    m.pbrExtras[0] = clamp01(src.baseColor[3]);
    m.pbrExtras[1] = clamp01(src.alphaCutoff);
    m.pbrExtras[2] = clamp01(src.transmission);
    m.pbrExtras[3] = static_cast<float>(alphaMode);
    return m;
}

void parseMaterial(const Json& mat, SourceMaterial& dst) {
    dst.name = mat.stringOr("name");
    const std::string alphaMode = mat.stringOr("alphaMode");
    if (!alphaMode.empty()) dst.alphaMode = alphaMode;
    dst.alphaCutoff = mat.floatOr("alphaCutoff", 0.5f);
    dst.doubleSided = mat.boolOr("doubleSided", false);
    std::vector<float> v;
    if (const Json* pbr = mat.object("pbrMetallicRoughness")) {
        if (pbr->floats("baseColorFactor", v) && v.size() >= 4) std::copy(v.begin(), v.begin() + 4, dst.baseColor);
        dst.metallic = pbr->floatOr("metallicFactor", 1.0f);
        dst.roughness = pbr->floatOr("roughnessFactor", 1.0f);
        parseBinding(pbr->object("baseColorTexture"), dst.baseColorTex);
        parseBinding(pbr->object("metallicRoughnessTexture"), dst.metalRoughTex);
    }
    if (const Json* t = mat.object("normalTexture")) {
        parseBinding(t, dst.normalTex);
        dst.normalScale = t->floatOr("scale", 1.0f);
    }
    if (const Json* t = mat.object("occlusionTexture")) {
        parseBinding(t, dst.occlusionTex);
        dst.occlusionStrength = t->floatOr("strength", 1.0f);
    }
    parseBinding(mat.object("emissiveTexture"), dst.emissiveTex);
    if (mat.floats("emissiveFactor", v) && v.size() >= 3) std::copy(v.begin(), v.begin() + 3, dst.emissive);
    const Json* ext = mat.object("extensions");
    if (!ext) return;
    if (const Json* t = ext->object("KHR_materials_transmission")) {
        dst.transmission = std::max(t->floatOr("transmissionFactor", 0.0f), 0.0f);
        parseBinding(t->object("transmissionTexture"), dst.transmissionTex);
    }
    if (const Json* vol = ext->object("KHR_materials_volume")) {
        dst.hasVolume = true;
        dst.thickness = std::max(vol->floatOr("thicknessFactor", 0.0f), 0.0f);
        if (vol->floats("attenuationColor", v) && v.size() >= 3) std::copy(v.begin(), v.begin() + 3, dst.attenuationColor);
        dst.attenuationDistance = vol->floatOr("attenuationDistance", std::numeric_limits<float>::infinity());
    }
    if (const Json* i = ext->object("KHR_materials_ior")) {
        dst.hasIor = true;
        dst.ior = i->floatOr("ior", 1.5f);
    }
    if (const Json* e = ext->object("KHR_materials_emissive_strength")) {
        dst.emissiveStrength = std::max(e->floatOr("emissiveStrength", 1.0f), 0.0f);
    }
}

// .glb container: 12-byte header, then (length, type, payload) chunks; JSON chunk required, BIN optional
bool splitGlb(const std::vector<uint8_t>& file, std::string& json, std::vector<uint8_t>& bin, std::string& error) {
    if (file.size() < 12) {
        error = "Invalid .glb header";
        return false;
    }
    uint32_t magic, version;
    std::memcpy(&magic, file.data(), 4);
    std::memcpy(&version, file.data() + 4, 4);
    if (magic != 0x46546C67u) {
        error = "Invalid .glb magic";
        return false;
    }
    if (version != 2u) {
        error = "Unsupported .glb version";
        return false;
    }
    bool haveJson = false;
    size_t at = 12;
    while (at + 8 <= file.size()) {
        uint32_t length, type;
        std::memcpy(&length, file.data() + at, 4);
        std::memcpy(&type, file.data() + at + 4, 4);
        at += 8;
        if (at + length > file.size()) {
            error = "Invalid .glb chunk length";
            return false;
        }
        if (type == 0x4E4F534Au) {
            json.assign(reinterpret_cast<const char*>(file.data() + at), length);
            haveJson = true;
        } else if (type == 0x004E4942u) {
            bin.assign(file.begin() + static_cast<long>(at), file.begin() + static_cast<long>(at + length));
        }
        at += length;
    }
    if (!haveJson) {
        error = "Missing JSON chunk in .glb";
        return false;
    }
    return true;
}

}  // namespace

bool LoadGltfScene(const std::string& path, SceneResources& resources, std::string& error, GltfCameraInfo* outCamera,
                   const GltfLoadOptions* optionsIn) {
    GltfLoadOptions options;
    if (optionsIn) options = *optionsIn;
    options.emissiveScale = std::max(options.emissiveScale, 0.0f);

    const fs::path gltfPath(path);
    std::vector<uint8_t> file;
    if (!readFileBytes(gltfPath, file, error)) return false;

    std::string jsonText;
    std::vector<uint8_t> binChunk;
    if (lowerAscii(gltfPath.extension().string()) == ".glb") {
        if (!splitGlb(file, jsonText, binChunk, error)) return false;
    } else {
        jsonText.assign(reinterpret_cast<const char*>(file.data()), file.size());
    }
    Json root;
    if (!JsonReader(jsonText.data(), jsonText.data() + jsonText.size()).parse(root) || root.kind != Json::Object) {
        error = "Failed to parse glTF JSON";
        return false;
    }

    Document doc;
    if (const Json* arr = root.array("buffers")) {
        doc.buffers.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const std::string uri = arr->items[i].stringOr("uri");
            if (!uri.empty()) {
                if (!decodeDataUri(uri, doc.buffers[i])) {
                    std::string readError;
                    if (!readFileBytes(gltfPath.parent_path() / uri, doc.buffers[i], readError)) {
                        error = readError;
                        return false;
                    }
                }
            } else if (!binChunk.empty()) {
                doc.buffers[i] = binChunk;
            } else {
                error = "glTF buffer missing uri and no .glb BIN chunk";
                return false;
            }
        }
    }
    if (const Json* arr = root.array("bufferViews")) {
        doc.views.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const Json& v = arr->items[i];
            doc.views[i].buffer = v.intOr("buffer", -1);
            doc.views[i].offset = static_cast<size_t>(std::max(v.intOr("byteOffset", 0), 0));
            doc.views[i].length = static_cast<size_t>(std::max(v.intOr("byteLength", 0), 0));
            doc.views[i].stride = static_cast<size_t>(std::max(v.intOr("byteStride", 0), 0));
        }
    }
    if (const Json* arr = root.array("accessors")) {
        doc.accessors.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const Json& a = arr->items[i];
            doc.accessors[i].view = a.intOr("bufferView", -1);
            doc.accessors[i].offset = static_cast<size_t>(std::max(a.intOr("byteOffset", 0), 0));
            doc.accessors[i].count = static_cast<size_t>(std::max(a.intOr("count", 0), 0));
            doc.accessors[i].componentType = a.intOr("componentType", 0);
            doc.accessors[i].type = a.stringOr("type");
            doc.accessors[i].normalized = a.boolOr("normalized", false);
        }
    }

    // ---- textures (GltfLoader.mm ResolveTextureIndex, 560-648): image bytes from a buffer view, a data URI or a file next to the
    // document; decoded here (PNG / baseline JPEG), converted to linear floats (base colour and emissive are sRGB-encoded) and
    // registered once per (image, sampler, encoding).  A texture that cannot be read is reported and left unbound.
    struct SamplerDef {
        uint32_t wrapS = 0, wrapT = 0, filter = 1;
    };
    struct TextureDef {
        int source = -1, sampler = -1;
    };
    struct ImageDef {
        int view = -1;
        std::string uri;
    };
    std::vector<SamplerDef> samplers;
    std::vector<TextureDef> textureDefs;
    std::vector<ImageDef> images;
    auto wrapOf = [](int gl) -> uint32_t { return gl == 33071 ? 1u : (gl == 33648 ? 2u : 0u); };
    if (const Json* arr = root.array("samplers")) {
        samplers.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            samplers[i].wrapS = wrapOf(arr->items[i].intOr("wrapS", 10497));
            samplers[i].wrapT = wrapOf(arr->items[i].intOr("wrapT", 10497));
            samplers[i].filter = arr->items[i].intOr("magFilter", 9729) == 9728 ? 0u : 1u;
        }
    }
    if (const Json* arr = root.array("textures")) {
        textureDefs.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            textureDefs[i].source = arr->items[i].intOr("source", -1);
            textureDefs[i].sampler = arr->items[i].intOr("sampler", -1);
        }
    }
    if (const Json* arr = root.array("images")) {
        images.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            images[i].view = arr->items[i].intOr("bufferView", -1);
            images[i].uri = arr->items[i].stringOr("uri");
        }
    }
    std::vector<std::pair<uint64_t, uint32_t>> textureCache;   // (texture index << 1 | srgb) -> registered texture
    std::vector<DecodedImage> decodedImages(images.size());
    std::vector<int> decodeState(images.size(), 0);            // 0 not tried, 1 ok, -1 failed
    auto resolveTexture = [&](int textureIndex, bool srgb) -> uint32_t {
        if (!options.loadTextures || textureIndex < 0 || textureIndex >= static_cast<int>(textureDefs.size())) return kNoTexture;
        const uint64_t key = (static_cast<uint64_t>(textureIndex) << 1) | (srgb ? 1u : 0u);
        for (const auto& e : textureCache) {
            if (e.first == key) return e.second;
        }
        const TextureDef& t = textureDefs[static_cast<size_t>(textureIndex)];
        if (t.source < 0 || t.source >= static_cast<int>(images.size())) return kNoTexture;
        const size_t im = static_cast<size_t>(t.source);
        if (decodeState[im] == 0) {
            std::vector<uint8_t> bytes;
            const uint8_t* data = nullptr;
            size_t size = 0;
            if (images[im].view >= 0 && images[im].view < static_cast<int>(doc.views.size())) {
                const BufferView& v = doc.views[static_cast<size_t>(images[im].view)];
                if (v.buffer >= 0 && v.buffer < static_cast<int>(doc.buffers.size()) && v.length <= doc.buffers[static_cast<size_t>(v.buffer)].size() && v.offset <= doc.buffers[static_cast<size_t>(v.buffer)].size() - v.length) {
                    data = doc.buffers[static_cast<size_t>(v.buffer)].data() + v.offset;
                    size = v.length;
                }
            } else if (!images[im].uri.empty()) {
                std::string readError;
                if (decodeDataUri(images[im].uri, bytes) || readFileBytes(gltfPath.parent_path() / images[im].uri, bytes, readError)) {
                    data = bytes.data();
                    size = bytes.size();
                }
            }
            std::string decodeError = "no image data";
            decodeState[im] = (data && DecodeImage(data, size, decodedImages[im], &decodeError)) ? 1 : -1;
            if (decodeState[im] < 0) std::fprintf(stderr, "[glTF] image %zu not loaded: %s\n", im, decodeError.c_str());
        }
        if (decodeState[im] < 0) return kNoTexture;
        SamplerDef sampler;
        if (t.sampler >= 0 && t.sampler < static_cast<int>(samplers.size())) sampler = samplers[static_cast<size_t>(t.sampler)];
        const DecodedImage& img = decodedImages[im];
        const uint32_t index = resources.addTexture(img.rgba.data(), img.width, img.height, srgb, sampler.wrapS, sampler.wrapT, sampler.filter);
        textureCache.push_back({key, index});
        return index;
    };

    // materials: a document without any still gets one default material (all factors 1, GltfLoader.mm:1061-1063)
    std::vector<SourceMaterial> materials;
    if (const Json* arr = root.array("materials")) {
        materials.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            if (arr->items[i].kind == Json::Object) parseMaterial(arr->items[i], materials[i]);
        }
    }
    if (materials.empty()) materials.emplace_back();
    std::vector<uint32_t> materialMap(materials.size(), 0u);
    for (size_t i = 0; i < materials.size(); ++i) {
        materials[i].disableOrm = containsNoCase(materials[i].name, "visor");   // quirk Q7
        PtrMaterial built = buildMaterial(materials[i], options);
        // texture slots (GltfLoader.mm:700-747): base colour and emissive are sRGB-encoded, the rest is linear data
        built.textureIndices0[0] = resolveTexture(materials[i].baseColorTex.index, !options.forceLinearBaseColor);
        built.textureIndices0[1] = resolveTexture(materials[i].metalRoughTex.index, false);
        built.textureIndices0[2] = resolveTexture(materials[i].normalTex.index, false);
        built.textureIndices0[3] = resolveTexture(materials[i].occlusionTex.index, false);
        built.textureIndices1[0] = resolveTexture(materials[i].emissiveTex.index, !options.forceLinearEmissive);
        built.textureIndices1[1] = resolveTexture(materials[i].transmissionTex.index, false);
        materialMap[i] = resources.addMaterialData(built, materials[i].name);
    }

    std::vector<MeshDef> meshes;
    if (const Json* arr = root.array("meshes")) {
        meshes.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const Json& mesh = arr->items[i];
            meshes[i].name = mesh.stringOr("name");
            const Json* prims = mesh.array("primitives");
            if (!prims) continue;
            for (const Json& prim : prims->items) {
                Primitive p;
                p.material = prim.intOr("material", -1);
                p.indices = prim.intOr("indices", -1);
                p.mode = prim.intOr("mode", 4);
                if (const Json* attrs = prim.object("attributes")) {
                    p.position = attrs->intOr("POSITION", -1);
                    p.normal = attrs->intOr("NORMAL", -1);
                    p.texcoord = attrs->intOr("TEXCOORD_0", -1);
                    p.texcoord1 = attrs->intOr("TEXCOORD_1", -1);
                    p.tangent = attrs->intOr("TANGENT", -1);
                }
                meshes[i].primitives.push_back(p);
            }
        }
    }

    std::vector<CameraDef> cameras;
    if (const Json* arr = root.array("cameras")) {
        cameras.resize(arr->items.size());
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const Json& cam = arr->items[i];
            if (cam.stringOr("type") == "perspective") {
                cameras[i].perspective = true;
                if (const Json* persp = cam.object("perspective")) cameras[i].yfov = persp->floatOr("yfov", 0.0f);
            }
        }
    }

    std::vector<NodeDef> nodes;
    if (const Json* arr = root.array("nodes")) {
        nodes.resize(arr->items.size());
        std::vector<float> v;
        for (size_t i = 0; i < arr->items.size(); ++i) {
            const Json& node = arr->items[i];
            NodeDef& n = nodes[i];
            n.name = node.stringOr("name");
            n.mesh = node.intOr("mesh", -1);
            n.camera = node.intOr("camera", -1);
            if (node.floats("translation", v) && v.size() >= 3) std::copy(v.begin(), v.begin() + 3, n.translation);
            if (node.floats("rotation", v) && v.size() >= 4) std::copy(v.begin(), v.begin() + 4, n.rotation);
            if (node.floats("scale", v) && v.size() >= 3) std::copy(v.begin(), v.begin() + 3, n.scale);
            if (node.floats("matrix", v) && v.size() >= 16) {
                n.hasMatrix = true;
                for (int c = 0; c < 4; ++c) n.matrix.columns[c] = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
            }
            if (const Json* kids = node.array("children")) {
                for (const Json& k : kids->items) {
                    if (k.kind == Json::Number) n.children.push_back(static_cast<int>(k.number));
                }
            }
        }
    }

    // roots: the default scene's node list; a document without scenes uses every node as a root
    std::vector<int> roots;
    const Json* scenes = root.array("scenes");
    const int sceneIndex = root.intOr("scene", 0);
    if (scenes && sceneIndex >= 0 && sceneIndex < static_cast<int>(scenes->items.size())) {
        if (const Json* list = scenes->items[static_cast<size_t>(sceneIndex)].array("nodes")) {
            for (const Json& k : list->items) {
                if (k.kind == Json::Number) roots.push_back(static_cast<int>(k.number));
            }
        }
    } else {
        for (size_t i = 0; i < nodes.size(); ++i) roots.push_back(static_cast<int>(i));
    }

    float3 boundsLo{0, 0, 0}, boundsHi{0, 0, 0};
    auto growSceneBounds = [&](const float3& lo, const float3& hi) {
        if (!outCamera) return;
        if (!outCamera->hasSceneBounds) {
            boundsLo = lo;
            boundsHi = hi;
            outCamera->hasSceneBounds = true;
        } else {
            // the reference re-derives a box from (centre, radius) before merging, which inflates it; keep that
            const float3 c = (boundsLo + boundsHi) * 0.5f;
            const float r = outCamera->sceneRadius;
            const float3 minC{std::min(c.x - r, lo.x), std::min(c.y - r, lo.y), std::min(c.z - r, lo.z)};
            const float3 maxC{std::max(c.x + r, hi.x), std::max(c.y + r, hi.y), std::max(c.z + r, hi.z)};
            boundsLo = minC;
            boundsHi = maxC;
        }
        outCamera->sceneRadius = length(boundsHi - boundsLo) * 0.5f;
    };

    auto loadPrimitive = [&](const Primitive& prim, const float4x4& localToWorld, const std::string& name) -> bool {
        if (prim.mode != 4) return true;   // triangles only
        if (prim.position < 0 || prim.position >= static_cast<int>(doc.accessors.size())) {
            error = "glTF primitive missing POSITION accessor";
            return false;
        }
        const Accessor& posAcc = doc.accessors[static_cast<size_t>(prim.position)];
        std::vector<float> positions;
        if (!doc.readFloats(posAcc, 3, positions)) {
            error = "Failed reading POSITION accessor";
            return false;
        }
        const size_t vertexCount = posAcc.count;
        if (vertexCount == 0) return true;

        if (outCamera) {
            float3 lo = transformPoint(localToWorld, {positions[0], positions[1], positions[2]}), hi = lo;
            for (size_t i = 1; i < vertexCount; ++i) {
                const float3 w = transformPoint(localToWorld, {positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]});
                lo = {std::min(lo.x, w.x), std::min(lo.y, w.y), std::min(lo.z, w.z)};
                hi = {std::max(hi.x, w.x), std::max(hi.y, w.y), std::max(hi.z, w.z)};
            }
            growSceneBounds(lo, hi);
        }

        std::vector<float> normals, uvs, uvs1, tangents;
        bool hasNormals = false, hasUvs = false, hasUvs1 = false, hasTangents = false;
        if (prim.texcoord1 >= 0 && prim.texcoord1 < static_cast<int>(doc.accessors.size())) {
            hasUvs1 = doc.readFloats(doc.accessors[static_cast<size_t>(prim.texcoord1)], 2, uvs1) && uvs1.size() >= vertexCount * 2;
        }
        if (prim.tangent >= 0 && prim.tangent < static_cast<int>(doc.accessors.size())) {
            hasTangents = doc.readFloats(doc.accessors[static_cast<size_t>(prim.tangent)], 4, tangents) && tangents.size() >= vertexCount * 4;
        }
        if (prim.normal >= 0 && prim.normal < static_cast<int>(doc.accessors.size())) {
            hasNormals = doc.readFloats(doc.accessors[static_cast<size_t>(prim.normal)], 3, normals) && normals.size() >= vertexCount * 3;
        }
        if (prim.texcoord >= 0 && prim.texcoord < static_cast<int>(doc.accessors.size())) {
            hasUvs = doc.readFloats(doc.accessors[static_cast<size_t>(prim.texcoord)], 2, uvs) && uvs.size() >= vertexCount * 2;
        }
        std::vector<uint32_t> indices;
        if (prim.indices >= 0 && prim.indices < static_cast<int>(doc.accessors.size())) {
            if (!doc.readIndices(doc.accessors[static_cast<size_t>(prim.indices)], indices)) {
                error = "Failed reading indices accessor";
                return false;
            }
        } else {
            indices.resize(vertexCount);
            for (size_t i = 0; i < vertexCount; ++i) indices[i] = static_cast<uint32_t>(i);
        }

        std::vector<SceneResources::MeshVertex> vertices(vertexCount);
        for (size_t i = 0; i < vertexCount; ++i) {
            SceneResources::MeshVertex& v = vertices[i];
            v.position = {positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]};
            if (hasNormals) v.normal = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
            if (hasUvs) v.uv = {uvs[2 * i], uvs[2 * i + 1]};
            if (hasUvs1) v.uv1 = {uvs1[2 * i], uvs1[2 * i + 1]};
            if (hasTangents) v.tangent = {tangents[4 * i], tangents[4 * i + 1], tangents[4 * i + 2], tangents[4 * i + 3] < 0.0f ? -1.0f : 1.0f};
        }
        if (!hasNormals) {
            // area-weighted vertex normals; vertices no valid triangle touches keep the default (0,1,0)
            std::vector<float3> accum(vertexCount, float3{0.0f, 0.0f, 0.0f});
            for (size_t i = 0; i + 2 < indices.size(); i += 3) {
                const uint32_t i0 = indices[i], i1 = indices[i + 1], i2 = indices[i + 2];
                if (i0 >= vertexCount || i1 >= vertexCount || i2 >= vertexCount) continue;
                const float3 n = cross(vertices[i1].position - vertices[i0].position, vertices[i2].position - vertices[i0].position);
                if (length(n) > 0.0f) {
                    accum[i0] = accum[i0] + n;
                    accum[i1] = accum[i1] + n;
                    accum[i2] = accum[i2] + n;
                }
            }
            for (size_t i = 0; i < vertexCount; ++i) {
                if (length(accum[i]) > 0.0f) vertices[i].normal = normalize(accum[i]);
            }
        }
        // a primitive with texture coordinates and no TANGENT gets MikkTSpace tangents, one vertex per triangle corner
        // (GltfLoader.mm:1447-1451 -> TangentGen.mm:181-230)
        bool generatedTangents = false;
        if (!hasTangents && hasUvs) {
            GenerateTangents(vertices, indices);
            generatedTangents = true;
        }
        uint32_t materialIndex = 0u;
        if (prim.material >= 0 && prim.material < static_cast<int>(materialMap.size())) materialIndex = materialMap[static_cast<size_t>(prim.material)];
        const uint32_t meshIndex = resources.addMesh(vertices.data(), static_cast<uint32_t>(vertices.size()), indices.data(),
                                                     static_cast<uint32_t>(indices.size()), localToWorld, materialIndex, name);
        // texture coordinates / tangents travel with the mesh (a mesh without texture coordinates has no tangents: the kernel then builds
        // an arbitrary frame, like the reference's when SceneVertex::tangent.w is 0 - shaders/pathtrace.metal:843-911)
        SceneResources::Mesh& stored = resources.meshAt(meshIndex);
        stored.hasUv0 = hasUvs;
        stored.hasUv1 = hasUvs1;
        stored.hasTangents = hasTangents || generatedTangents;
        return true;
    };

    // depth-first walk; a node reachable twice is instanced twice, cycles are cut by the depth limit
    std::function<bool(int, const float4x4&, const std::string&, int)> visit;
    visit = [&](int index, const float4x4& parent, const std::string& prefix, int depth) -> bool {
        if (index < 0 || index >= static_cast<int>(nodes.size()) || depth > 256) return true;
        const NodeDef& node = nodes[static_cast<size_t>(index)];
        const float4x4 world = mul(parent, localMatrix(node));
        std::string nodeName = prefix;
        if (!node.name.empty()) nodeName += (nodeName.empty() ? "" : "/") + node.name;

        if (node.camera >= 0 && node.camera < static_cast<int>(cameras.size()) && outCamera && !outCamera->valid) {
            const CameraDef& cam = cameras[static_cast<size_t>(node.camera)];
            outCamera->valid = true;
            outCamera->hasPerspective = cam.perspective;
            outCamera->yfov = cam.yfov;
            outCamera->position = transformPoint(world, {0.0f, 0.0f, 0.0f});
            outCamera->forward = normalize(transformDirection(world, {0.0f, 0.0f, -1.0f}));
        }
        if (node.mesh >= 0 && node.mesh < static_cast<int>(meshes.size())) {
            const MeshDef& mesh = meshes[static_cast<size_t>(node.mesh)];
            for (size_t p = 0; p < mesh.primitives.size(); ++p) {
                std::string meshName = nodeName;
                if (!mesh.name.empty()) meshName += (meshName.empty() ? "" : "/") + mesh.name;
                if (mesh.primitives.size() > 1) meshName += ".prim" + std::to_string(p);
                if (!loadPrimitive(mesh.primitives[p], world, meshName)) return false;
            }
        }
        for (int child : node.children) {
            if (!visit(child, world, nodeName, depth + 1)) return false;
        }
        return true;
    };
    for (int r : roots) {
        if (!visit(r, float4x4::identity(), "", 0)) return false;
    }
    return true;
}

}  // namespace ptr
