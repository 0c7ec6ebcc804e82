// Headless backend plug-in boundary.  Same shape as the reference's
// include/headless/IHeadlessRenderer.h:17-52 (HeadlessScene / HeadlessCamera / HeadlessRenderOutput /
// IHeadlessRenderer::render) so that the CLI code path stays `renderer->render(...)`.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "render_settings.h"
#include "scene_resources.h"

namespace ptr {

enum class HeadlessBackend { Hip = 0 };

struct HeadlessScene {
    std::string source;
    bool isPath = false;
    const SceneResources* resources = nullptr;
};

struct HeadlessCamera {  // passed for interface parity; backends rebuild the camera from settings (quirk Q5)
    float3 target{0.0f, 0.0f, 0.0f};
    float distance = 0.0f;
    float yaw = 0.0f;
    float pitch = 0.0f;
    float verticalFov = 0.0f;
    float defocusAngle = 0.0f;
    float focusDistance = 0.0f;
};

struct HeadlessRenderOutput {
    std::vector<float> linearRGB;  // W*H*3, row 0 = top, linear, mean over spp
    uint32_t width = 0;
    uint32_t height = 0;
    uint32_t samples = 0;
    double totalSeconds = 0.0;
    double avgMsPerSample = 0.0;
};

class IHeadlessRenderer {
public:
    virtual ~IHeadlessRenderer() = default;
    virtual bool render(const HeadlessScene& scene, const HeadlessCamera& camera, const RenderSettings& settings,
                        uint32_t sppTotal, bool verbose, HeadlessRenderOutput& out, std::string& error) = 0;
};

// MI355X backend: wraps the C-ABI (ptr_render) behind the reference's interface.
class HipHeadlessRenderer : public IHeadlessRenderer {
public:
    bool render(const HeadlessScene& scene, const HeadlessCamera& camera, const RenderSettings& settings,
                uint32_t sppTotal, bool verbose, HeadlessRenderOutput& out, std::string& error) override;
    const PtrRenderStats& lastStats() const { return m_stats; }

private:
    PtrRenderStats m_stats{};
};

// RenderSettings -> POD settings of the C-ABI.
void FillPtrSettings(const RenderSettings& settings, PtrSettings& out);

}  // namespace ptr
