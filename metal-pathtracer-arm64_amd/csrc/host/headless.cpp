#include "headless.h"

#include <algorithm>

namespace ptr {

void FillPtrSettings(const RenderSettings& s, PtrSettings& o) {
    o = PtrSettings{};
    o.width = s.renderWidth > 0 ? s.renderWidth : 1280u;   // EmbreeHeadlessRenderer.mm:2462-2463
    o.height = s.renderHeight > 0 ? s.renderHeight : 720u;
    o.maxDepth = s.maxDepth;
    o.seed = s.fixedRngSeed;
    o.enableRussianRoulette = s.enableRussianRoulette ? 1u : 0u;
    o.enableSpecularNee = s.enableSpecularNee ? 1u : 0u;
    o.enableMnee = s.enableMnee ? 1u : 0u;
    o.enableMneeSecondary = s.enableMneeSecondary ? 1u : 0u;
    o.cameraTarget[0] = s.cameraTarget.x;
    o.cameraTarget[1] = s.cameraTarget.y;
    o.cameraTarget[2] = s.cameraTarget.z;
    o.cameraDistance = s.cameraDistance;
    o.cameraYaw = s.cameraYaw;
    o.cameraPitch = s.cameraPitch;
    o.cameraVerticalFov = s.cameraVerticalFov;
    o.cameraDefocusAngle = s.cameraDefocusAngle;
    o.cameraFocusDistance = s.cameraFocusDistance;
    o.backgroundMode = static_cast<uint32_t>(s.backgroundMode);
    o.backgroundColor[0] = s.backgroundColor.x;
    o.backgroundColor[1] = s.backgroundColor.y;
    o.backgroundColor[2] = s.backgroundColor.z;
    o.environmentRotation = s.environmentRotation;
    o.environmentIntensity = s.environmentIntensity;
    o.fireflyClampEnabled = s.fireflyClampEnabled ? 1u : 0u;
    o.fireflyClampFactor = s.fireflyClampFactor;
    o.fireflyClampFloor = s.fireflyClampFloor;
    o.throughputClamp = s.throughputClamp;
    o.specularTailClampBase = s.specularTailClampBase;
    o.specularTailClampRoughnessScale = s.specularTailClampRoughnessScale;
    o.minSpecularPdf = s.minSpecularPdf;
    o.fireflyClampMaxContribution = s.fireflyClampMaxContribution;
    o.emissionScale = 1.0f;
    o.metalSemantics = s.metalSemantics;
    o.sssMode = static_cast<uint32_t>(s.sssMode);
    o.sssMaxSteps = s.sssMaxSteps;
    o.debugShadowSlack = 0.0f;
}

bool HipHeadlessRenderer::render(const HeadlessScene& scene, const HeadlessCamera&, const RenderSettings& settings,
                                 uint32_t sppTotal, bool verbose, HeadlessRenderOutput& out, std::string& error) {
    if (!scene.resources) {
        error = "HIP backend requires scene resources";
        return false;
    }
    PtrSceneDesc desc;
    scene.resources->fillSceneDesc(desc);
    PtrSettings ps;
    FillPtrSettings(settings, ps);

    const uint32_t spp = std::max<uint32_t>(1u, sppTotal);
    out.linearRGB.assign(static_cast<size_t>(ps.width) * ps.height * 3u, 0.0f);
    char err[512] = {0};
    m_stats = PtrRenderStats{};
    m_aovAlbedo.clear();
    m_aovNormal.clear();
    if (m_devices != 1) {
        // the frame in interleaved bands over several devices of the node, gathered on the first one
        if (ptr_render_multi(&desc, &ps, spp, m_devices, verbose ? 1 : 0, out.linearRGB.data(), &m_stats, err, sizeof(err)) != 0) {
            error = err[0] ? err : "HIP render failed";
            return false;
        }
    } else if (!m_captureAovs) {
        if (ptr_render(&desc, &ps, spp, verbose ? 1 : 0, out.linearRGB.data(), &m_stats, err, sizeof(err)) != 0) {
            error = err[0] ? err : "HIP render failed";
            return false;
        }
    }
    if (m_captureAovs) {
        // one upload serves the frame (single device) and the first-hit feature buffers
        PtrDeviceScene* ds = nullptr;
        if (ptr_scene_upload(&desc, 0, &ds, err, sizeof(err)) != 0) {
            error = err[0] ? err : "HIP scene upload failed";
            return false;
        }
        bool ok = true;
        if (m_devices == 1) {
            const uint32_t bands = ptr_part_band_count(ps.height, 0, 1);
            std::vector<float> banded(static_cast<size_t>(bands) * PTR_BAND_ROWS * ps.width * 3u);
            ok = ptr_render_bands(ds, &ps, spp, 0, 1, banded.data(), 0, &m_stats, err, sizeof(err)) == 0;
            if (ok) std::copy(banded.begin(), banded.begin() + static_cast<std::ptrdiff_t>(out.linearRGB.size()), out.linearRGB.begin());
        }
        if (ok) {
            m_aovAlbedo.assign(static_cast<size_t>(ps.width) * ps.height * 4u, 0.0f);
            m_aovNormal.assign(static_cast<size_t>(ps.width) * ps.height * 4u, 0.0f);
            ok = ptr_render_aovs(ds, &ps, 0u, m_aovAlbedo.data(), m_aovNormal.data(), err, sizeof(err)) == 0;
        }
        ptr_scene_release(ds);
        if (!ok) {
            error = err[0] ? err : "HIP render failed";
            return false;
        }
    }
    out.width = ps.width;
    out.height = ps.height;
    out.samples = spp;
    out.totalSeconds = m_stats.totalSeconds;
    out.avgMsPerSample = m_stats.avgMsPerSample;
    return true;
}

}  // namespace ptr
