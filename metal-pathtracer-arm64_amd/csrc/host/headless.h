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

// What the caller hands over: where the scene came from and the parsed CPU-side arrays (the HIP backend, like the
// reference's Embree backend, consumes `resources`; `source` / `isPath` are kept for messages).
struct HeadlessScene {
    const SceneResources* resources = nullptr;
    std::string source;
    bool isPath = false;
};

// Orbit camera as the CLI resolved it.  Passed for interface parity only: both reference backends rebuild the camera
// from the settings (quirk Q5), and so does this one.
struct HeadlessCamera {
    float3 target{0.0f, 0.0f, 0.0f};
    float distance = 0.0f, yaw = 0.0f, pitch = 0.0f;
    float verticalFov = 0.0f, defocusAngle = 0.0f, focusDistance = 0.0f;
};

// Result: linear RGB, width*height*3 floats, row 0 = top, mean over the samples; timing of the integrate phase.
struct HeadlessRenderOutput {
    uint32_t width = 0, height = 0, samples = 0;
    double totalSeconds = 0.0, avgMsPerSample = 0.0;
    std::vector<float> linearRGB;
};

// The plug-in interface: false + message on failure, no exceptions across it, called once from the main thread.
class IHeadlessRenderer {
public:
    virtual bool render(const HeadlessScene& scene, const HeadlessCamera& camera, const RenderSettings& settings,
                        uint32_t sppTotal, bool verbose, HeadlessRenderOutput& out, std::string& error) = 0;
    virtual ~IHeadlessRenderer() = default;
};

// MI355X backend: wraps the C-ABI (ptr_render) behind the reference's interface.
class HipHeadlessRenderer : public IHeadlessRenderer {
public:
    bool render(const HeadlessScene& scene, const HeadlessCamera& camera, const RenderSettings& settings,
                uint32_t sppTotal, bool verbose, HeadlessRenderOutput& out, std::string& error) override;
    const PtrRenderStats& lastStats() const { return m_stats; }
    // devices of this node to spread the frame over (1 = the first device only, 0 = all visible): --devices of the CLI
    void setDeviceCount(int n) { m_devices = n; }
    // also keep the first-hit feature buffers of the frame (albedo rgb | hit, shading normal * 0.5 + 0.5 | distance; width*height*4
    // floats each) - what the reference hands to its denoiser (shaders/pathtrace.metal:6424-6435, 9813-9815): --aovExr of the CLI
    void setCaptureAovs(bool on) { m_captureAovs = on; }
    const std::vector<float>& aovAlbedo() const { return m_aovAlbedo; }
    const std::vector<float>& aovNormal() const { return m_aovNormal; }

private:
    PtrRenderStats m_stats{};
    int m_devices = 1;
    bool m_captureAovs = false;
    std::vector<float> m_aovAlbedo, m_aovNormal;
};

// RenderSettings -> POD settings of the C-ABI.
void FillPtrSettings(const RenderSettings& settings, PtrSettings& out);

}  // namespace ptr
