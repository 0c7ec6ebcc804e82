// Image output for the headless CLI: PFM / PPM / scanline EXR (byte-compatible with the reference) and PNG;
// reference: src/renderer/ImageWriter.mm (WritePPM 164-191, WritePFM 193-214, WriteScanlineEXR 239-464, WritePNG 480-565).
#pragma once

#include <cstdint>
#include <string>

namespace ptr {

enum class ImageFileFormat { EXR, PNG, PFM, PPM };

struct TonemapSettings {
    uint32_t tonemapMode = 1;  // 1=Linear, 2=ACES, 3=Reinhard, 4=Hable
    uint32_t acesVariant = 0;  // 0=fitted, 1=simple
    float exposure = 0.0f;
    float reinhardWhitePoint = 1.5f;
};

bool ParseImageFileFormat(const std::string& value, ImageFileFormat& outFormat);
const char* FormatExtension(ImageFileFormat format);

// linearRGB: width*height*3 floats, row 0 = top.
bool WriteImage(const std::string& path, ImageFileFormat format, const float* linearRGB, uint32_t width,
                uint32_t height, const TonemapSettings& tonemap, std::string* errorMessage = nullptr);

// RGBA EXR with optional colorspace string attribute (what main_headless.mm:568-583 writes for Embree).
bool WriteExrRgba(const std::string& path, const float* rgba, uint32_t width, uint32_t height,
                  const char* colorspace, std::string* errorMessage = nullptr);

// RGBA + a planar SAMPLES channel (per-pixel sample counts), ImageWriter.mm:657-684; falls back to WriteExrRgba
// when sampleCount is null.
bool WriteExrMultilayer(const std::string& path, const float* rgba, uint32_t width, uint32_t height, const float* sampleCount,
                        const char* colorspace, std::string* errorMessage = nullptr);

// Beauty + denoiser feature layers in one scanline EXR: R G B, albedo.R/G/B, normal.X/Y/Z (unit vectors, 0 on a miss), depth.Z
// (hit distance, 0 on a miss).  albedoRgba / normalRgba: the buffers of ptr_render_aovs (width*height*4 floats each).
bool WriteExrAovs(const std::string& path, const float* rgb, const float* albedoRgba, const float* normalRgba, uint32_t width, uint32_t height,
                  std::string* errorMessage = nullptr);

// 8-bit tonemapped RGB (shared by PPM writer and tests).
void TonemapToLdr(const float* linearRGB, uint32_t pixelCount, const TonemapSettings& tonemap, uint8_t* outRgb8);

}  // namespace ptr
