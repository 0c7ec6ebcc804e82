// SceneManager — `.scene` text format loader (directives camera / renderer / background / material /
// sphere / box / rectangle|rect / mesh).  Grammar and error strings follow the reference's
// src/renderer/SceneManager.mm:791-2633.
#pragma once

#include <istream>
#include <string>
#include <unordered_map>
#include <vector>

#include "render_settings.h"
#include "scene_resources.h"

namespace ptr {

class SceneManager {
public:
    // sceneDirectory: base for relative `mesh path=` / `background env=` values.  Empty -> the reference's
    // rule (<cwd>/assets when it exists, else the process CWD, SceneManager.mm:570-615, 2401-2407); files
    // not found there are additionally looked up next to the .scene file (convenience extension).
    explicit SceneManager(std::string sceneDirectory = {});

    // Scene catalogue of the scene directory (SceneManager.mm:646-675, 724-789, 2635-2663): every *.scene file,
    // identifier = file stem, display name = text of a leading `#` comment line (else the identifier), sorted
    // by display name.
    struct SceneInfo {
        std::string identifier, displayName, filePath;
    };
    bool refresh(std::string* errorMessage = nullptr);
    const std::vector<SceneInfo>& scenes() const { return m_scenes; }
    const SceneInfo* findScene(const std::string& identifier) const;
    bool loadScene(const std::string& identifier, SceneResources& resources, RenderSettings& inOutSettings,
                   std::string* errorMessage = nullptr);

    bool loadSceneFromPath(const std::string& path, SceneResources& resources, RenderSettings& inOutSettings,
                           std::string* errorMessage = nullptr);
    bool parseScene(std::istream& stream, SceneResources& resources, RenderSettings& inOutSettings,
                    std::string& errorMessage) const;

    const std::string& sceneDirectory() const { return m_sceneDirectory; }

    using Tokens = std::unordered_map<std::string, std::string>;
    static Tokens tokenize(const std::string& line);
    static std::string trim(const std::string& value);
    static bool parseFloat(const std::string& value, float& out);
    static bool parseUInt(const std::string& value, uint32_t& out);
    static bool parseFloat3(const std::string& value, float3& out);
    static bool parseFloatRange(const std::string& value, float& outMin, float& outMax, bool& outIsFixed);
    static bool parseMaterialType(const std::string& value, MaterialType& out);

private:
    bool parseCamera(const Tokens& tokens, RenderSettings& s, std::string& err) const;
    bool parseRenderer(const Tokens& tokens, RenderSettings& s, std::string& err) const;
    bool parseBackground(const Tokens& tokens, RenderSettings& s, std::string& err) const;
    bool parseMaterial(const Tokens& tokens, SceneResources& r, std::string& err,
                       std::unordered_map<std::string, uint32_t>& materialIndicesByName) const;
    bool parseSphere(const Tokens& tokens, SceneResources& r, std::string& err) const;
    bool parseBox(const Tokens& tokens, SceneResources& r, std::string& err) const;
    bool parseRectangle(const Tokens& tokens, SceneResources& r, std::string& err) const;
    bool parseMesh(const Tokens& tokens, SceneResources& r, std::string& err, RenderSettings& s,
                   bool allowEmbeddedCameraOverride,
                   const std::unordered_map<std::string, uint32_t>& materialIndicesByName) const;
    bool resolveAssetPath(const std::string& value, bool hdrSubdir, std::string& outPath) const;

    static std::string readDisplayName(const std::string& filePath);

    std::string m_sceneDirectory;
    mutable std::string m_sceneFileDirectory;
    std::vector<SceneInfo> m_scenes;
    std::unordered_map<std::string, size_t> m_sceneIndexById;
};

}  // namespace ptr
