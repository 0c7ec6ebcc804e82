// TEST INFRASTRUCTURE (oracle/_ref): a driver of ours around the reference's own mesh parsers, external/tinyobjloader/tiny_obj_loader.h
// and external/tinyply/tinyply.h, compiled where they lie under /root/reference by oracle/Makefile.  Each entry parses a file with the
// library, configured as the reference configures it, and turns the library's output into vertex / index arrays the way the
// reference's loader does (src/renderer/SceneManager.mm:96-209 OBJ: one vertex per distinct (position, normal, texcoord) index triple
// in order of first use; 223-518 PLY: vertices as stored, polygons fanned from their first corner).  Pins
// csrc/host/mesh_loaders.cpp from outside (tests/test_reference_loaders.py).
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#define TINYPLY_IMPLEMENTATION
#include "tinyply.h"

namespace {

struct Loaded {
    std::vector<float> positions, normals, uvs;   // 3 / 3 / 2 per vertex
    std::vector<uint32_t> indices;
    std::string error;
};

template <typename T>
void copyOut(const std::vector<T>& v, T* dst) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(T));
}

}  // namespace

extern "C" {

void* ref_mesh_open_obj(const char* path) {
    auto out = std::make_unique<Loaded>();
    tinyobj::ObjReaderConfig config;
    config.triangulate = true;
    config.vertex_color = false;
    tinyobj::ObjReader reader;
    if (!reader.ParseFromFile(path, config)) {
        out->error = reader.Error().empty() ? "parse failed" : reader.Error();
        return out.release();
    }
    const auto& attrib = reader.GetAttrib();
    std::map<std::tuple<int, int, int>, uint32_t> seen;
    for (const auto& shape : reader.GetShapes()) {
        for (const auto& index : shape.mesh.indices) {
            const auto key = std::make_tuple(index.vertex_index, index.normal_index, index.texcoord_index);
            auto found = seen.find(key);
            if (found == seen.end()) {
                const uint32_t id = static_cast<uint32_t>(out->positions.size() / 3);
                found = seen.emplace(key, id).first;
                for (int k = 0; k < 3; ++k) out->positions.push_back(attrib.vertices[static_cast<size_t>(index.vertex_index) * 3 + k]);
                for (int k = 0; k < 3; ++k) {   // a vertex without a normal keeps the loader's default (0, 1, 0)
                    out->normals.push_back(index.normal_index >= 0 ? attrib.normals[static_cast<size_t>(index.normal_index) * 3 + k] : (k == 1 ? 1.0f : 0.0f));
                }
                for (int k = 0; k < 2; ++k) out->uvs.push_back(index.texcoord_index >= 0 ? attrib.texcoords[static_cast<size_t>(index.texcoord_index) * 2 + k] : 0.0f);
            }
            out->indices.push_back(found->second);
        }
    }
    return out.release();
}

void* ref_mesh_open_ply(const char* path) {
    auto out = std::make_unique<Loaded>();
    try {
        std::ifstream stream(path, std::ios::binary);
        if (!stream.is_open()) throw std::runtime_error("cannot open");
        tinyply::PlyFile ply;
        ply.parse_header(stream);
        auto request = [&](const char* element, std::vector<std::string> names, uint32_t hint = 0) -> std::shared_ptr<tinyply::PlyData> {
            try {
                return ply.request_properties_from_element(element, names, hint);
            } catch (const std::exception&) {
                return nullptr;
            }
        };
        auto positions = request("vertex", {"x", "y", "z"});
        auto normals = request("vertex", {"nx", "ny", "nz"});
        auto uvs = request("vertex", {"u", "v"});
        if (!uvs) uvs = request("vertex", {"s", "t"});
        if (!uvs) uvs = request("vertex", {"texture_u", "texture_v"});
        auto faces = request("face", {"vertex_indices"}, 3);
        if (!faces) faces = request("face", {"vertex_index"}, 3);
        ply.read(stream);
        if (!positions || !faces) throw std::runtime_error("no positions / faces");
        auto floats = [&](const std::shared_ptr<tinyply::PlyData>& d, size_t per, std::vector<float>& dst) {
            dst.assign(d->count * per, 0.0f);
            if (d->t == tinyply::Type::FLOAT32) {
                std::memcpy(dst.data(), d->buffer.get(), dst.size() * 4);
            } else if (d->t == tinyply::Type::FLOAT64) {
                const double* src = reinterpret_cast<const double*>(d->buffer.get());
                for (size_t i = 0; i < dst.size(); ++i) dst[i] = static_cast<float>(src[i]);
            } else {
                throw std::runtime_error("unsupported float type");
            }
        };
        floats(positions, 3, out->positions);
        out->normals.assign(positions->count * 3, 0.0f);   // PLY vertices start without a normal (SceneManager.mm:299)
        if (normals) floats(normals, 3, out->normals);
        out->uvs.assign(positions->count * 2, 0.0f);
        if (uvs) floats(uvs, 2, out->uvs);
        auto fan = [&](auto tag) {
            using I = decltype(tag);
            const I* src = reinterpret_cast<const I*>(faces->buffer.get());
            const size_t values = faces->buffer.size_bytes() / sizeof(I), per = values / faces->count;
            for (size_t f = 0; f < faces->count; ++f) {
                for (size_t k = 1; k + 1 < per; ++k) {
                    out->indices.push_back(static_cast<uint32_t>(src[f * per]));
                    out->indices.push_back(static_cast<uint32_t>(src[f * per + k]));
                    out->indices.push_back(static_cast<uint32_t>(src[f * per + k + 1]));
                }
            }
        };
        switch (faces->t) {
            case tinyply::Type::UINT8: fan(uint8_t{}); break;
            case tinyply::Type::INT8: fan(int8_t{}); break;
            case tinyply::Type::UINT16: fan(uint16_t{}); break;
            case tinyply::Type::INT16: fan(int16_t{}); break;
            case tinyply::Type::UINT32: fan(uint32_t{}); break;
            case tinyply::Type::INT32: fan(int32_t{}); break;
            default: throw std::runtime_error("unsupported index type");
        }
    } catch (const std::exception& e) {
        out->error = e.what();
    }
    return out.release();
}

// counts: [0] vertices, [1] indices; returns 0 when the file parsed
int ref_mesh_counts(void* handle, uint64_t counts[2], char* err, uint64_t err_cap) {
    const Loaded* m = static_cast<const Loaded*>(handle);
    counts[0] = m->positions.size() / 3;
    counts[1] = m->indices.size();
    if (!m->error.empty() && err && err_cap) std::snprintf(err, err_cap, "%s", m->error.c_str());
    return m->error.empty() ? 0 : 1;
}

void ref_mesh_copy(void* handle, float* positions, float* normals, float* uvs, uint32_t* indices) {
    const Loaded* m = static_cast<const Loaded*>(handle);
    copyOut(m->positions, positions);
    copyOut(m->normals, normals);
    copyOut(m->uvs, uvs);
    copyOut(m->indices, indices);
}

void ref_mesh_close(void* handle) { delete static_cast<Loaded*>(handle); }

}  // extern "C"
