#include "scene_manager.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <limits>
#include <sstream>

#include "gltf_loader.h"
#include "mesh_loaders.h"

namespace fs = std::filesystem;

namespace ptr {
namespace {

constexpr float kPi = 3.14159265358979323846f;
const char* const kKeywordToken = "__keyword";

// Aluminium-like defaults for the car-paint base conductor (SceneManager.mm:40-41).
const float3 kDefaultCarpaintBaseEta{1.3456f, 0.9652f, 0.6172f};
const float3 kDefaultCarpaintBaseK{7.4746f, 6.3995f, 5.3031f};

std::string lowered(const std::string& s) {
    std::string r;
    r.reserve(s.size());
    for (char c : s) r.push_back(static_cast<char>(std::tolower(static_cast<unsigned char>(c))));
    return r;
}

inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }
inline float3 clamp01(const float3& v) { return {clampf(v.x, 0, 1), clampf(v.y, 0, 1), clampf(v.z, 0, 1)}; }
inline float3 maxZero(const float3& v) { return {std::max(v.x, 0.0f), std::max(v.y, 0.0f), std::max(v.z, 0.0f)}; }

// Small readers that report "<directive> <key> expects ..." errors like the reference does.
struct TokenReader {
    const SceneManager::Tokens& tokens;
    const char* directive;
    std::string& err;

    const std::string* find(const char* key) const {
        auto it = tokens.find(key);
        return it == tokens.end() ? nullptr : &it->second;
    }
    // returns false only on a malformed value; `present` tells whether the key existed
    bool f(const char* key, float& out, bool* present = nullptr, const char* what = "a float") const {
        const std::string* v = find(key);
        if (present) *present = v != nullptr;
        if (!v) return true;
        if (!SceneManager::parseFloat(*v, out)) {
            err = std::string(directive) + " " + key + " expects " + what;
            return false;
        }
        return true;
    }
    bool u(const char* key, uint32_t& out, bool* present = nullptr, const char* what = "an integer") const {
        const std::string* v = find(key);
        if (present) *present = v != nullptr;
        if (!v) return true;
        if (!SceneManager::parseUInt(*v, out)) {
            err = std::string(directive) + " " + key + " expects " + what;
            return false;
        }
        return true;
    }
    bool f3(const char* key, float3& out, bool* present = nullptr, const char* what = "three floats") const {
        const std::string* v = find(key);
        if (present) *present = v != nullptr;
        if (!v) return true;
        if (!SceneManager::parseFloat3(*v, out)) {
            err = std::string(directive) + " " + key + " expects " + what;
            return false;
        }
        return true;
    }
    bool flag01(const char* key, bool& out) const {
        uint32_t v = 0;
        bool present = false;
        if (!u(key, v, &present, "0 or 1")) return false;
        if (present) out = (v != 0);
        return true;
    }
};

bool parseOnOff(const std::string& value, bool& out) {
    const std::string l = lowered(value);
    if (l == "on" || l == "true" || l == "1") {
        out = true;
        return true;
    }
    if (l == "off" || l == "false" || l == "0") {
        out = false;
        return true;
    }
    return false;
}

float4x4 scaleMatrix(const float3& s) {
    float4x4 m = float4x4::identity();
    m.columns[0].x = s.x;
    m.columns[1].y = s.y;
    m.columns[2].z = s.z;
    return m;
}

float4x4 translationMatrix(const float3& t) {
    float4x4 m = float4x4::identity();
    m.columns[3] = float4(t, 1.0f);
    return m;
}

// R = Rz * Ry * Rx, angles in degrees (SceneManager.mm:534-559).
float4x4 rotationMatrix(const float3& deg) {
    const float rx = deg.x * (kPi / 180.0f), ry = deg.y * (kPi / 180.0f), rz = deg.z * (kPi / 180.0f);
    const float sx = std::sin(rx), cx = std::cos(rx);
    const float sy = std::sin(ry), cy = std::cos(ry);
    const float sz = std::sin(rz), cz = std::cos(rz);
    float4x4 X = float4x4::identity(), Y = float4x4::identity(), Z = float4x4::identity();
    X.columns[1] = {0.0f, cx, sx, 0.0f};
    X.columns[2] = {0.0f, -sx, cx, 0.0f};
    Y.columns[0] = {cy, 0.0f, -sy, 0.0f};
    Y.columns[2] = {sy, 0.0f, cy, 0.0f};
    Z.columns[0] = {cz, sz, 0.0f, 0.0f};
    Z.columns[1] = {-sz, cz, 0.0f, 0.0f};
    return mul(mul(Z, Y), X);
}

float4x4 composeTransform(const float3& t, const float3& rotDeg, const float3& s) {
    return mul(translationMatrix(t), mul(rotationMatrix(rotDeg), scaleMatrix(s)));
}

}  // namespace

SceneManager::SceneManager(std::string sceneDirectory) : m_sceneDirectory(std::move(sceneDirectory)) {
    if (m_sceneDirectory.empty()) {
        std::error_code ec;
        const fs::path cwd = fs::current_path(ec);
        if (!ec && fs::is_directory(cwd / "assets", ec)) {
            m_sceneDirectory = (cwd / "assets").string();
        }
    }
    refresh(nullptr);
}

bool SceneManager::refresh(std::string* errorMessage) {
    m_scenes.clear();
    m_sceneIndexById.clear();
    if (m_sceneDirectory.empty()) return true;
    std::error_code ec;
    const fs::path dir(m_sceneDirectory);
    if (!fs::exists(dir, ec) || !fs::is_directory(dir, ec)) {
        if (errorMessage) *errorMessage = "Scene directory is not accessible: " + m_sceneDirectory;
        return false;
    }
    for (fs::directory_iterator it(dir, ec), end; !ec && it != end; it.increment(ec)) {
        if (!it->is_regular_file() || it->path().extension() != ".scene") continue;
        SceneInfo info;
        std::error_code absEc;
        const fs::path absolute = fs::absolute(it->path(), absEc);
        info.filePath = absEc ? it->path().string() : absolute.string();
        info.identifier = it->path().stem().string();
        info.displayName = readDisplayName(info.filePath);
        if (info.displayName.empty()) info.displayName = info.identifier;
        m_scenes.push_back(std::move(info));
    }
    if (ec) {
        if (errorMessage) *errorMessage = "Failed iterating scene directory: " + ec.message();
        return false;
    }
    std::sort(m_scenes.begin(), m_scenes.end(), [](const SceneInfo& a, const SceneInfo& b) { return a.displayName < b.displayName; });
    for (size_t i = 0; i < m_scenes.size(); ++i) m_sceneIndexById[m_scenes[i].identifier] = i;
    return true;
}

std::string SceneManager::readDisplayName(const std::string& filePath) {
    std::ifstream stream(filePath);
    std::string line;
    while (stream.is_open() && std::getline(stream, line)) {
        std::string trimmed = trim(line);
        if (trimmed.empty()) continue;
        if (trimmed.front() == '#') return trim(trimmed.substr(1));
        break;
    }
    return {};
}

const SceneManager::SceneInfo* SceneManager::findScene(const std::string& identifier) const {
    const auto it = m_sceneIndexById.find(identifier);
    return it == m_sceneIndexById.end() ? nullptr : &m_scenes[it->second];
}

bool SceneManager::loadScene(const std::string& identifier, SceneResources& resources, RenderSettings& inOutSettings,
                             std::string* errorMessage) {
    const SceneInfo* info = findScene(identifier);
    if (!info) {
        if (errorMessage) *errorMessage = "Unknown scene identifier: " + identifier;
        return false;
    }
    return loadSceneFromPath(info->filePath, resources, inOutSettings, errorMessage);
}

bool SceneManager::loadSceneFromPath(const std::string& path, SceneResources& resources,
                                     RenderSettings& inOutSettings, std::string* errorMessage) {
    std::ifstream stream(path);
    if (!stream.is_open()) {
        if (errorMessage) *errorMessage = "Failed to open scene file: " + path;
        return false;
    }
    {
        std::error_code ec;
        fs::path abs = fs::absolute(fs::path(path), ec);
        m_sceneFileDirectory = ec ? std::string() : abs.parent_path().string();
    }

    resources.clear();
    // A scene file starts from defaults for camera lens, background, clamps and size, but inherits the
    // rest of the caller's settings (SceneManager.mm:689-709).
    RenderSettings parsed = inOutSettings;
    const RenderSettings d{};
    parsed.cameraVerticalFov = d.cameraVerticalFov;
    parsed.cameraDefocusAngle = d.cameraDefocusAngle;
    parsed.cameraFocusDistance = d.cameraFocusDistance;
    parsed.backgroundMode = d.backgroundMode;
    parsed.backgroundColor = d.backgroundColor;
    parsed.environmentMapPath = d.environmentMapPath;
    parsed.environmentRotation = d.environmentRotation;
    parsed.environmentIntensity = d.environmentIntensity;
    parsed.fireflyClampEnabled = d.fireflyClampEnabled;
    parsed.fireflyClampFactor = d.fireflyClampFactor;
    parsed.fireflyClampFloor = d.fireflyClampFloor;
    parsed.throughputClamp = d.throughputClamp;
    parsed.specularTailClampBase = d.specularTailClampBase;
    parsed.specularTailClampRoughnessScale = d.specularTailClampRoughnessScale;
    parsed.minSpecularPdf = d.minSpecularPdf;
    parsed.renderWidth = d.renderWidth;
    parsed.renderHeight = d.renderHeight;
    parsed.enableSoftwareRayTracing = d.enableSoftwareRayTracing;

    std::string parseError;
    if (!parseScene(stream, resources, parsed, parseError)) {
        resources.clear();
        if (errorMessage) *errorMessage = "Failed parsing scene '" + path + "': " + parseError;
        return false;
    }
    if (!parsed.environmentMapPath.empty()) {
        std::string envError;
        if (!resources.loadEnvironmentMap(parsed.environmentMapPath, envError)) {
            // The Embree backend silently renders without the map when it cannot be decoded
            // (EmbreeHeadlessRenderer.mm:2467-2471); keep going, background falls back to the sky.
        }
    }
    inOutSettings = parsed;
    return true;
}

bool SceneManager::parseScene(std::istream& stream, SceneResources& resources, RenderSettings& settings,
                              std::string& errorMessage) const {
    bool sawCamera = false;
    std::unordered_map<std::string, uint32_t> materialsByName;

    auto dispatch = [&](const std::string& content, size_t startLine) -> bool {
        const Tokens tokens = tokenize(content);
        auto kw = tokens.find(kKeywordToken);
        if (kw == tokens.end()) return true;
        const std::string& keyword = kw->second;
        std::string localError;
        bool ok;
        if (keyword == "camera") {
            ok = parseCamera(tokens, settings, localError);
            sawCamera = sawCamera || ok;
        } else if (keyword == "renderer") {
            ok = parseRenderer(tokens, settings, localError);
        } else if (keyword == "background") {
            ok = parseBackground(tokens, settings, localError);
        } else if (keyword == "material") {
            ok = parseMaterial(tokens, resources, localError, materialsByName);
        } else if (keyword == "sphere") {
            ok = parseSphere(tokens, resources, localError);
        } else if (keyword == "box") {
            ok = parseBox(tokens, resources, localError);
        } else if (keyword == "rectangle" || keyword == "rect") {
            ok = parseRectangle(tokens, resources, localError);
        } else if (keyword == "mesh") {
            ok = parseMesh(tokens, resources, localError, settings, !sawCamera, materialsByName);
        } else {
            return true;  // unknown directives are ignored
        }
        if (!ok) {
            errorMessage = "line " + std::to_string(startLine) + ": " + localError;
        }
        return ok;
    };

    // Logical lines: '#' and blank lines terminate a pending continuation; trailing '\' joins lines.
    std::string line, pending;
    size_t lineNumber = 0, pendingStart = 0;
    auto flushPending = [&](size_t fallbackLine) -> bool {
        if (pending.empty()) return true;
        const bool ok = dispatch(pending, pendingStart == 0 ? fallbackLine : pendingStart);
        pending.clear();
        pendingStart = 0;
        return ok;
    };
    while (std::getline(stream, line)) {
        ++lineNumber;
        std::string text = trim(line);
        if (text.empty() || text[0] == '#') {
            if (!flushPending(lineNumber)) return false;
            continue;
        }
        bool continued = false;
        if (text.back() == '\\') {
            continued = true;
            text.pop_back();
            text = trim(text);
        }
        if (!text.empty()) {
            if (pending.empty()) {
                pending = text;
                pendingStart = lineNumber;
            } else {
                pending += " " + text;
            }
        }
        if (!continued && !flushPending(lineNumber)) return false;
    }
    return flushPending(lineNumber);
}

SceneManager::Tokens SceneManager::tokenize(const std::string& line) {
    Tokens tokens;
    std::istringstream stream(line);
    std::string word;
    if (!(stream >> word)) return tokens;
    tokens.emplace(kKeywordToken, word);
    while (stream >> word) {
        const size_t eq = word.find('=');
        if (eq == std::string::npos) continue;  // bare words carry no meaning
        tokens[word.substr(0, eq)] = word.substr(eq + 1);
    }
    return tokens;
}

std::string SceneManager::trim(const std::string& value) {
    size_t b = 0, e = value.size();
    while (b < e && std::isspace(static_cast<unsigned char>(value[b]))) ++b;
    while (e > b && std::isspace(static_cast<unsigned char>(value[e - 1]))) --e;
    return value.substr(b, e - b);
}

bool SceneManager::parseFloat(const std::string& value, float& out) {
    const std::string t = trim(value);
    if (t.empty()) return false;
    char* end = nullptr;
    errno = 0;
    const float v = std::strtof(t.c_str(), &end);
    if (errno != 0 || end == t.c_str() || *end != '\0') return false;
    out = v;
    return true;
}

bool SceneManager::parseUInt(const std::string& value, uint32_t& out) {
    const std::string t = trim(value);
    if (t.empty()) return false;
    char* end = nullptr;
    errno = 0;
    const unsigned long v = std::strtoul(t.c_str(), &end, 10);
    if (errno != 0 || end == t.c_str() || *end != '\0') return false;
    if (v > std::numeric_limits<uint32_t>::max()) return false;
    out = static_cast<uint32_t>(v);
    return true;
}

bool SceneManager::parseFloat3(const std::string& value, float3& out) {
    float c[3] = {0, 0, 0};
    int n = 0;
    std::istringstream stream(value);
    std::string part;
    while (std::getline(stream, part, ',')) {
        if (n >= 3 || !parseFloat(part, c[n])) return false;
        ++n;
    }
    if (n != 3) return false;
    out = {c[0], c[1], c[2]};
    return true;
}

bool SceneManager::parseFloatRange(const std::string& value, float& outMin, float& outMax, bool& outIsFixed) {
    const std::string t = trim(value);
    if (t.empty()) return false;
    const size_t comma = t.find(',');
    if (comma == std::string::npos) {
        if (!parseFloat(t, outMin)) return false;
        outMax = outMin;
        outIsFixed = true;
        return true;
    }
    if (!parseFloat(t.substr(0, comma), outMin) || !parseFloat(t.substr(comma + 1), outMax)) return false;
    if (outMin > outMax) std::swap(outMin, outMax);
    outIsFixed = std::fabs(outMax - outMin) < 1e-6f;
    return true;
}

bool SceneManager::parseMaterialType(const std::string& value, MaterialType& out) {
    static const struct {
        const char* name;
        MaterialType type;
    } kNames[] = {
        {"lambert", MaterialType::Lambertian},   {"lambertian", MaterialType::Lambertian},
        {"metal", MaterialType::Metal},          {"metallic", MaterialType::Metal},
        {"dielectric", MaterialType::Dielectric}, {"glass", MaterialType::Dielectric},
        {"diffuse_light", MaterialType::DiffuseLight}, {"light", MaterialType::DiffuseLight},
        {"emissive", MaterialType::DiffuseLight}, {"plastic", MaterialType::Plastic},
        {"sss", MaterialType::Subsurface},       {"subsurface", MaterialType::Subsurface},
        {"carpaint", MaterialType::CarPaint},    {"car_paint", MaterialType::CarPaint},
        {"automotive", MaterialType::CarPaint},
    };
    const std::string l = lowered(value);
    for (const auto& e : kNames) {
        if (l == e.name) {
            out = e.type;
            return true;
        }
    }
    return false;
}

bool SceneManager::parseCamera(const Tokens& tokens, RenderSettings& s, std::string& err) const {
    TokenReader rd{tokens, "camera", err};
    bool present = false;
    float v = 0.0f;
    if (!rd.f3("target", s.cameraTarget, nullptr, "three comma-separated floats")) return false;
    if (!rd.f("distance", v, &present)) return false;
    if (present) s.cameraDistance = std::max(v, 0.0f);
    if (!rd.f("yaw", s.cameraYaw, nullptr, "a float (radians)")) return false;
    if (!rd.f("pitch", s.cameraPitch, nullptr, "a float (radians)")) return false;
    if (!rd.f("vfov", s.cameraVerticalFov, nullptr, "a float (degrees)")) return false;
    if (!rd.f("defocusAngle", v, &present, "a float (degrees)")) return false;
    if (present) s.cameraDefocusAngle = std::max(v, 0.0f);
    if (!rd.f("focusDist", s.cameraFocusDistance)) return false;
    return true;
}

bool SceneManager::parseRenderer(const Tokens& tokens, RenderSettings& s, std::string& err) const {
    TokenReader rd{tokens, "renderer", err};
    bool present = false;
    uint32_t u = 0;
    float f = 0.0f;

    if (!rd.u("samplesPerFrame", u, &present)) return false;
    if (present) s.samplesPerFrame = std::max<uint32_t>(1u, u);
    if (!rd.u("width", u, &present)) return false;
    if (present) s.renderWidth = std::max<uint32_t>(u, 8u);
    if (!rd.u("height", u, &present)) return false;
    if (present) s.renderHeight = std::max<uint32_t>(u, 8u);
    if (!rd.u("maxDepth", s.maxDepth)) return false;
    if (!rd.u("tonemap", u, &present)) return false;
    if (present) s.tonemapMode = std::max<uint32_t>(1u, std::min<uint32_t>(u, 4u));
    if (!rd.f("exposure", s.exposure)) return false;
    if (!rd.f("envRotation", f, &present, "a float (degrees)")) return false;
    if (present) s.environmentRotation = f * (kPi / 180.0f);
    if (!rd.f("envIntensity", f, &present)) return false;
    if (present) s.environmentIntensity = std::max(f, 0.0f);
    if (!rd.f("reinhardWhite", s.reinhardWhitePoint)) return false;
    if (!rd.u("seed", s.fixedRngSeed)) return false;
    if (!rd.flag01("russianRoulette", s.enableRussianRoulette)) return false;
    if (!rd.u("acesVariant", s.acesVariant)) return false;
    for (const char* key : {"enableSoftwareRayTracing", "softwareRayTracing", "forceSoftwareBvh"}) {
        if (!rd.flag01(key, s.enableSoftwareRayTracing)) return false;
    }
    if (const std::string* v = rd.find("sss")) {
        const std::string l = lowered(*v);
        if (l == "off" || l == "disabled" || l == "0") {
            s.sssMode = RenderSettings::SssMode::Off;
        } else if (l == "separable" || l == "diffusion" || l == "approx") {
            s.sssMode = RenderSettings::SssMode::Separable;
        } else if (l == "randomwalk" || l == "random_walk" || l == "random-walk") {
            s.sssMode = RenderSettings::SssMode::RandomWalk;
        } else {
            err = "renderer sss expects off, separable, or randomwalk";
            return false;
        }
    }
    if (!rd.u("sssMaxSteps", u, &present)) return false;
    if (present) s.sssMaxSteps = std::max<uint32_t>(1u, u);
    if (!rd.flag01("fireflyClampEnabled", s.fireflyClampEnabled)) return false;

    struct {
        const char* key;
        float* dst;
    } nonNegative[] = {
        {"fireflyClampFactor", &s.fireflyClampFactor},
        {"fireflyClampFloor", &s.fireflyClampFloor},
        {"throughputClamp", &s.throughputClamp},
        {"specularTailClampBase", &s.specularTailClampBase},
        {"specularTailClampRoughnessScale", &s.specularTailClampRoughnessScale},
        {"minSpecularPdf", &s.minSpecularPdf},
        {"fireflyClampMaxContribution", &s.fireflyClampMaxContribution},
        {"gltfEmissiveScale", &s.gltfEmissiveScale},
    };
    for (auto& e : nonNegative) {
        if (!rd.f(e.key, f, &present)) return false;
        if (present) *e.dst = std::max(f, 0.0f);
    }
    if (!rd.flag01("enableSpecularNee", s.enableSpecularNee)) return false;
    if (!rd.flag01("enableMnee", s.enableMnee)) return false;
    if (!rd.flag01("enableMneeSecondary", s.enableMneeSecondary)) return false;
    if (!rd.flag01("gltfViewerCompatibilityMode", s.gltfViewerCompatibilityMode)) return false;
    if (!rd.flag01("gltfCompat", s.gltfViewerCompatibilityMode)) return false;
    if (!rd.flag01("gltfThinWalledFallback", s.gltfThinWalledFallback)) return false;
    if (!rd.flag01("gltfThinFallback", s.gltfThinWalledFallback)) return false;
    // bloom*, debug* and the other gltfCompat* keys only drive the GUI / Metal texture path: accepted, ignored.
    return true;
}

bool SceneManager::resolveAssetPath(const std::string& value, bool hdrSubdir, std::string& outPath) const {
    fs::path p(value);
    std::vector<fs::path> candidates;
    if (p.is_relative()) {
        auto addBase = [&](const std::string& base) {
            if (base.empty()) return;
            if (hdrSubdir && !p.has_parent_path()) {
                candidates.push_back(fs::path(base) / "HDR" / p);
            }
            candidates.push_back(fs::path(base) / p);
        };
        if (!m_sceneDirectory.empty()) {
            addBase(m_sceneDirectory);
        } else {
            std::error_code ec;
            addBase(fs::current_path(ec).string());
        }
        addBase(m_sceneFileDirectory);
    } else {
        candidates.push_back(p);
    }
    for (const fs::path& c : candidates) {
        std::error_code ec;
        const fs::path canonical = fs::weakly_canonical(c, ec);
        if (!ec && fs::exists(canonical, ec)) {
            outPath = canonical.string();
            return true;
        }
    }
    outPath = candidates.empty() ? value : candidates.front().string();
    return false;
}

bool SceneManager::parseBackground(const Tokens& tokens, RenderSettings& s, std::string& err) const {
    TokenReader rd{tokens, "background", err};
    const std::string* solid = rd.find("solid");
    const std::string* env = rd.find("env");
    if (solid && env) {
        err = "background cannot specify both solid and env";
        return false;
    }
    if (solid) {
        float3 color;
        if (!parseFloat3(*solid, color)) {
            err = "background solid expects three floats";
            return false;
        }
        s.backgroundMode = RenderSettings::BackgroundMode::Solid;
        s.backgroundColor = color;
        s.environmentMapPath.clear();
        return true;
    }
    if (env) {
        std::string resolved;
        if (!resolveAssetPath(*env, /*hdrSubdir=*/true, resolved)) {
            err = "background env map not found: " + resolved;
            return false;
        }
        s.backgroundMode = RenderSettings::BackgroundMode::Environment;
        s.backgroundColor = {0.0f, 0.0f, 0.0f};
        s.environmentMapPath = resolved;
        return true;
    }
    s.backgroundMode = RenderSettings::BackgroundMode::Gradient;
    s.backgroundColor = {0.0f, 0.0f, 0.0f};
    s.environmentMapPath.clear();
    return true;
}

bool SceneManager::parseMaterial(const Tokens& tokens, SceneResources& resources, std::string& err,
                                 std::unordered_map<std::string, uint32_t>& materialsByName) const {
    TokenReader rd{tokens, "material", err};
    const std::string* typeToken = rd.find("type");
    if (!typeToken) {
        err = "material requires a type token";
        return false;
    }
    MaterialParams m;
    if (!parseMaterialType(*typeToken, m.type)) {
        err = "material type is not recognized";
        return false;
    }
    const bool isPlastic = m.type == MaterialType::Plastic;
    const bool isSss = m.type == MaterialType::Subsurface;
    const bool isCarPaint = m.type == MaterialType::CarPaint;
    bool present = false;

    // colour: first of base / albedo / color
    for (const char* key : {"base", "albedo", "color"}) {
        if (rd.find(key)) {
            if (!rd.f3(key, m.baseColor)) return false;
            break;
        }
    }

    float roughness = 0.0f, fuzz = 0.0f;
    bool roughnessExplicit = false;
    if (!rd.f("roughness", roughness, &roughnessExplicit)) return false;
    if (roughnessExplicit) roughness = clampf(roughness, 0.0f, 1.0f);
    if (!rd.f("fuzz", fuzz, &present)) return false;
    if (present) fuzz = clampf(fuzz, 0.0f, 1.0f);
    if (!roughnessExplicit) roughness = fuzz;

    float ior = 1.5f;
    bool iorExplicit = false;
    if (!rd.f("ior", ior, &iorExplicit)) return false;

    if (rd.find("emit")) {
        if (!rd.f3("emit", m.emission)) return false;
    } else if (!rd.f3("emission", m.emission)) {
        return false;
    }
    if (rd.find("emitEnv")) {
        if (!rd.flag01("emitEnv", m.emissionUsesEnvironment)) return false;
    } else if (!rd.flag01("envPortal", m.emissionUsesEnvironment)) {
        return false;
    }
    if (m.type == MaterialType::DiffuseLight) {
        roughness = 0.0f;
        ior = 1.0f;
    }
    if (const std::string* name = rd.find("name")) m.name = *name;

    for (const char* key : {"thin", "thinWalled", "thinDielectric"}) {
        if (const std::string* v = rd.find(key)) {
            if (!parseOnOff(*v, m.thinDielectric)) {
                err = std::string("material ") + key + " expects on/off";
                return false;
            }
            break;
        }
    }

    // --- clear coat (plastic / sss / car paint) ---
    m.coatRoughness = (isPlastic || isSss) ? 0.05f : (isCarPaint ? 0.04f : 0.0f);
    float coatIor = 1.5f;

    // --- car paint ---
    float flakeDensity = 0.0f;
    if (isCarPaint) {
        if (!rd.f("baseMetallic", m.carpaintBaseMetallic, &present)) return false;
        if (present) m.carpaintBaseMetallic = clampf(m.carpaintBaseMetallic, 0.0f, 1.0f);

        float baseRoughness = roughnessExplicit ? roughness : 0.2f;
        float explicitBase = 0.0f;
        if (!rd.f("baseRoughness", explicitBase, &present)) return false;
        if (present) baseRoughness = clampf(explicitBase, 0.0f, 1.0f);
        m.carpaintBaseRoughness = baseRoughness;

        if (!rd.f("flakeDensity", flakeDensity, &present)) return false;
        flakeDensity = present ? std::max(flakeDensity, 0.0f) : 2000000.0f;
        if (!rd.f("flakeRoughness", m.carpaintFlakeRoughness, &present)) return false;
        m.carpaintFlakeRoughness = present ? clampf(m.carpaintFlakeRoughness, 0.0f, 1.0f) : 0.15f;
        if (!rd.f("flakeAnisotropy", m.carpaintFlakeAnisotropy, &present)) return false;
        m.carpaintFlakeAnisotropy = present ? clampf(m.carpaintFlakeAnisotropy, -0.99f, 0.99f) : 0.3f;
        if (!rd.f("flakeScale", m.carpaintFlakeScale, &present)) return false;
        m.carpaintFlakeScale = present ? std::max(m.carpaintFlakeScale, 1.0e-4f) : 0.5f;
        m.carpaintFlakeNormalStrength = 0.35f;
        if (!rd.f("flakeNormalStrength", m.carpaintFlakeNormalStrength, &present)) return false;
        m.carpaintFlakeNormalStrength = clampf(m.carpaintFlakeNormalStrength, 0.0f, 1.0f);
        if (!rd.f("flakeReflectanceScale", m.carpaintFlakeReflectanceScale, &present)) return false;
        m.carpaintFlakeReflectanceScale = clampf(m.carpaintFlakeReflectanceScale, 0.0f, 1.0f);
        if (!rd.f3("baseTint", m.carpaintBaseTint, &present)) return false;
        m.carpaintBaseTint = clamp01(m.carpaintBaseTint);

        m.carpaintBaseEta = kDefaultCarpaintBaseEta;
        m.carpaintBaseK = kDefaultCarpaintBaseK;
        bool etaExplicit = false, kExplicit = false;
        if (!rd.f3("baseEta", m.carpaintBaseEta, &etaExplicit)) return false;
        if (etaExplicit) m.carpaintBaseEta = maxZero(m.carpaintBaseEta);
        if (!rd.f3("baseK", m.carpaintBaseK, &kExplicit)) return false;
        if (kExplicit) m.carpaintBaseK = maxZero(m.carpaintBaseK);

        roughness = baseRoughness;
        m.carpaintHasBaseConductor = etaExplicit || kExplicit || (m.carpaintBaseMetallic > 1.0e-4f);
        m.carpaintFlakeSampleWeight = clampf(flakeDensity * 1.0e-7f, 0.0f, 0.6f);
    }

    if (isPlastic || isSss || isCarPaint) {
        if (!rd.f("coatRoughness", m.coatRoughness, &present)) return false;
        if (present) m.coatRoughness = clampf(m.coatRoughness, 0.0f, 1.0f);
        if (!rd.f("coatThickness", m.coatThickness, &present)) return false;
        if (present) m.coatThickness = std::max(m.coatThickness, 0.0f);
        if (!rd.f3("coatTint", m.coatTint, &present)) return false;
        if (present) m.coatTint = clamp01(m.coatTint);
        if (!rd.f3("coatAbsorption", m.coatAbsorption, &present)) return false;
        if (present) m.coatAbsorption = maxZero(m.coatAbsorption);
    }
    if (!rd.f("coatIOR", coatIor)) return false;
    m.coatIor = coatIor;
    if (isPlastic && !iorExplicit) ior = coatIor;
    if (isCarPaint && !iorExplicit) ior = 1.5f;

    if (isSss) {
        if (const std::string* v = rd.find("coat")) {
            if (!parseOnOff(*v, m.sssCoatEnabled)) {
                err = "material coat expects on/off";
                return false;
            }
        }
    }

    if (m.type == MaterialType::Metal) {
        bool etaPresent = false, kPresent = false;
        if (!rd.f3("eta", m.conductorEta, &etaPresent)) return false;
        if (!rd.f3("k", m.conductorK, &kPresent)) return false;
        m.hasConductorParameters = etaPresent || kPresent;
    }

    if (isSss) {
        m.sssMeanFreePath = 1.0f;
        if (const std::string* v = rd.find("method")) {
            const std::string l = lowered(*v);
            if (l == "separable" || l == "diffusion") {
                m.sssMethod = 0u;
            } else if (l == "randomwalk" || l == "random_walk") {
                m.sssMethod = 1u;
            } else {
                err = "material method for sss must be separable or randomwalk";
                return false;
            }
        }
        if (!rd.f("mfp", m.sssMeanFreePath)) return false;
        if (!rd.f("g", m.sssAnisotropy, &present)) return false;
        if (present) m.sssAnisotropy = clampf(m.sssAnisotropy, -0.99f, 0.99f);
        bool sa = false, ss = false;
        if (!rd.f3("sigma_a", m.sssSigmaA, &sa)) return false;
        if (!rd.f3("sigma_s", m.sssSigmaS, &ss)) return false;
        if (sa != ss) {
            err = "material sigma_a and sigma_s must both be provided together";
            return false;
        }
        m.sssSigmaA = maxZero(m.sssSigmaA);
        m.sssSigmaS = maxZero(m.sssSigmaS);
        m.sssSigmaOverride = sa && ss;
        m.sssMeanFreePath = std::max(m.sssMeanFreePath, 1.0e-4f);
    }

    // glass absorption: sigmaA=r,g,b  or  absorption=r,g,b thickness=t
    if (rd.find("sigmaA")) {
        if (!rd.f3("sigmaA", m.dielectricSigmaA)) return false;
        m.dielectricSigmaA = maxZero(m.dielectricSigmaA);
    } else if (rd.find("absorption") && rd.find("thickness")) {
        float3 absorption;
        float thickness = 0.0f;
        if (!rd.f3("absorption", absorption)) return false;
        if (!rd.f("thickness", thickness)) return false;
        const float denom = std::max(thickness, 1.0e-6f);
        m.dielectricSigmaA = maxZero({absorption.x / denom, absorption.y / denom, absorption.z / denom});
    }

    m.roughness = roughness;
    m.indexOfRefraction = ior;
    const uint32_t index = resources.addMaterial(m);
    if (!m.name.empty()) materialsByName[m.name] = index;
    return true;
}

bool SceneManager::parseSphere(const Tokens& tokens, SceneResources& resources, std::string& err) const {
    TokenReader rd{tokens, "sphere", err};
    if (!rd.find("center") || !rd.find("radius") || !rd.find("material")) {
        err = "sphere requires center, radius, and material tokens";
        return false;
    }
    float3 center;
    float radius = 0.0f;
    uint32_t material = 0;
    if (!rd.f3("center", center)) return false;
    if (!rd.f("radius", radius)) return false;
    if (!rd.u("material", material, nullptr, "an integer index")) return false;
    if (material >= resources.materialCount()) {
        err = "sphere references material index that has not been defined yet";
        return false;
    }
    resources.addSphere(center, radius, material);
    return true;
}

bool SceneManager::parseBox(const Tokens& tokens, SceneResources& resources, std::string& err) const {
    TokenReader rd{tokens, "box", err};
    if (!rd.find("min") || !rd.find("max") || !rd.find("material")) {
        err = "box requires min, max, and material tokens";
        return false;
    }
    float3 lo, hi;
    uint32_t material = 0;
    if (!rd.f3("min", lo) || !rd.f3("max", hi)) return false;
    if (!rd.u("material", material, nullptr, "an integer index")) return false;
    if (material >= resources.materialCount()) {
        err = "box references material index that has not been defined yet";
        return false;
    }
    bool includeBottom = true, twoSided = false;
    if (!rd.flag01("includeBottom", includeBottom) || !rd.flag01("twoSided", twoSided)) return false;

    float3 translate;
    float rotateY = 0.0f;
    bool hasTranslate = false, hasRotate = false;
    if (!rd.f3("translate", translate, &hasTranslate)) return false;
    if (!rd.f("rotateY", rotateY, &hasRotate, "a float (degrees)")) return false;
    if (!hasTranslate && !hasRotate) {
        resources.addBox(lo, hi, material, includeBottom, twoSided);
        return true;
    }
    const float radians = rotateY * (kPi / 180.0f);
    const float c = std::cos(radians), s = std::sin(radians);
    float4x4 rotation = float4x4::identity();
    rotation.columns[0] = {c, 0.0f, -s, 0.0f};
    rotation.columns[2] = {s, 0.0f, c, 0.0f};
    resources.addBoxTransformed(lo, hi, material, mul(translationMatrix(translate), rotation), includeBottom, twoSided);
    return true;
}

bool SceneManager::parseRectangle(const Tokens& tokens, SceneResources& resources, std::string& err) const {
    TokenReader rd{tokens, "rectangle", err};
    uint32_t material = 0;
    if (!rd.find("material")) {
        err = "rectangle requires a material token";
        return false;
    }
    if (!rd.u("material", material, nullptr, "an integer index")) return false;
    if (material >= resources.materialCount()) {
        err = "rectangle references material index that has not been defined yet";
        return false;
    }
    const char* labels[3] = {"x", "y", "z"};
    float lo[3], hi[3];
    bool fixed[3];
    uint32_t fixedCount = 0, normalAxis = 0;
    for (uint32_t a = 0; a < 3; ++a) {
        const std::string* v = rd.find(labels[a]);
        if (!v) {
            err = std::string("rectangle requires ") + labels[a] + " token";
            return false;
        }
        if (!parseFloatRange(*v, lo[a], hi[a], fixed[a])) {
            err = std::string("rectangle ") + labels[a] + " expects either a single value or a min,max range";
            return false;
        }
        if (fixed[a]) {
            normalAxis = a;
            ++fixedCount;
        }
    }
    if (fixedCount != 1) {
        err = "rectangle requires exactly one axis to be fixed to a single value";
        return false;
    }
    bool normalPositive = true;
    float normalValue = 1.0f;
    bool present = false;
    if (!rd.f("normal", normalValue, &present)) return false;
    if (present) normalPositive = normalValue >= 0.0f;
    bool twoSided = false;
    if (!rd.flag01("twoSided", twoSided)) return false;
    resources.addRectangle({lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}, normalAxis, normalPositive, twoSided, material);
    return true;
}

bool SceneManager::parseMesh(const Tokens& tokens, SceneResources& resources, std::string& err,
                             RenderSettings& settings, bool allowEmbeddedCameraOverride,
                             const std::unordered_map<std::string, uint32_t>& materialsByName) const {
    TokenReader rd{tokens, "mesh", err};
    std::string meshName;
    if (const std::string* n = rd.find("name")) meshName = *n;
    const std::string typeLower = rd.find("type") ? lowered(*rd.find("type")) : std::string();
    const bool isPlane = (typeLower == "plane" || typeLower == "quad");

    std::string meshPath;
    if (!isPlane) {
        const std::string* pathValue = rd.find("path") ? rd.find("path") : rd.find("file");
        if (!pathValue) {
            err = "mesh requires path or file token";
            return false;
        }
        if (!resolveAssetPath(*pathValue, /*hdrSubdir=*/false, meshPath)) {
            err = "mesh file not found: " + meshPath;
            return false;
        }
    }

    float3 translation, rotationDeg, scale{1.0f, 1.0f, 1.0f};
    if (rd.find("translate")) {
        if (!rd.f3("translate", translation)) return false;
    } else if (!rd.f3("position", translation)) {
        return false;
    }
    if (!rd.f3("rotate", rotationDeg, nullptr, "three floats (degrees)")) return false;
    if (const std::string* sv = rd.find("scale")) {
        if (!parseFloat3(*sv, scale)) {
            float uniform = 1.0f;
            if (!parseFloat(*sv, uniform)) {
                err = "mesh scale expects one float or three floats";
                return false;
            }
            scale = {uniform, uniform, uniform};
        }
    }
    const float4x4 localToWorld = composeTransform(translation, rotationDeg, scale);

    std::string ext;
    if (!isPlane) ext = lowered(fs::path(meshPath).extension().string());

    if (ext == ".gltf" || ext == ".glb") {
        GltfLoadOptions options;
        options.emissiveScale = std::max(settings.gltfEmissiveScale, 0.0f);
        GltfCameraInfo camera;
        const size_t firstMesh = resources.meshes().size();
        std::string gltfError;
        if (!LoadGltfScene(meshPath, resources, gltfError, &camera, &options)) {
            err = gltfError.empty() ? "Failed to load glTF scene" : gltfError;
            return false;
        }
        for (size_t i = firstMesh; i < resources.meshes().size(); ++i) {
            const uint32_t mi = static_cast<uint32_t>(i);
            resources.setMeshTransform(mi, mul(localToWorld, resources.meshTransform(mi)));
        }
        // An embedded perspective camera applies only when no `camera` directive came first
        // (SceneManager.mm:2528-2546): orbit parameters are derived from its pose.
        if (allowEmbeddedCameraOverride && camera.valid && camera.hasPerspective && camera.yfov > 0.0f) {
            const float distance = camera.hasSceneBounds ? std::max(camera.sceneRadius * 2.0f, 0.1f) : 1.0f;
            const float3 target = camera.position + normalize(camera.forward) * distance;
            const float3 offset = camera.position - target;
            settings.cameraTarget = target;
            settings.cameraDistance = distance;
            settings.cameraYaw = std::atan2(offset.z, offset.x);
            settings.cameraPitch = std::atan2(offset.y, std::sqrt(offset.x * offset.x + offset.z * offset.z));
            settings.cameraVerticalFov = camera.yfov * (180.0f / kPi);
            settings.cameraDefocusAngle = 0.0f;
            settings.cameraFocusDistance = distance;
        }
        return true;
    }

    const std::string* materialToken = rd.find("material");
    if (!materialToken) {
        err = "mesh requires material token";
        return false;
    }
    uint32_t material = 0;
    if (!parseUInt(*materialToken, material)) {
        auto it = materialsByName.find(*materialToken);
        if (it == materialsByName.end()) {
            err = "mesh material expects an index or known material name";
            return false;
        }
        material = it->second;
    }
    if (material >= resources.materialCount()) {
        err = "mesh references material index that has not been defined yet";
        return false;
    }

    LoadedMeshData data;
    if (isPlane) {
        // unit quad in the XZ plane, +Y normal (SceneManager.mm:2566-2583)
        const float px[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
        const float pz[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
        const float uu[4] = {0.0f, 1.0f, 1.0f, 0.0f};
        const float vv[4] = {0.0f, 0.0f, 1.0f, 1.0f};
        data.vertices.resize(4);
        for (int i = 0; i < 4; ++i) {
            data.vertices[i].position = {px[i], 0.0f, pz[i]};
            data.vertices[i].normal = {0.0f, 1.0f, 0.0f};
            data.vertices[i].uv.x = uu[i];
            data.vertices[i].uv.y = vv[i];
        }
        data.indices = {0, 1, 2, 0, 2, 3};
    } else {
        std::string loadError;
        bool loaded;
        if (ext == ".obj") {
            loaded = LoadObjMesh(meshPath, data, loadError);
        } else if (ext == ".ply") {
            loaded = LoadPlyMesh(meshPath, data, loadError);
        } else {
            err = "mesh format not supported: " + fs::path(meshPath).extension().string();
            return false;
        }
        if (!loaded) {
            err = loadError;
            return false;
        }
    }
    if (data.indices.size() % 3 != 0) {
        err = "mesh loader produced a non-triangle index buffer";
        return false;
    }
    if (data.vertices.empty() || data.indices.empty()) {
        err = "mesh contains no renderable geometry";
        return false;
    }
    resources.addMesh(data.vertices.data(), static_cast<uint32_t>(data.vertices.size()), data.indices.data(),
                      static_cast<uint32_t>(data.indices.size()), localToWorld, material, std::move(meshName));
    return true;
}

}  // namespace ptr
