// Environment-map decoding to linear RGBA32F, row 0 = top.
// The reference goes through MTKTextureLoader (EmbreeHeadlessRenderer.mm:1920-2004), which is an Apple
// framework; here Radiance RGBE (.hdr) and PFM (.pfm) are decoded directly.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ptr {

bool LoadHdrImageRgba(const std::string& path, std::vector<float>& rgba, uint32_t& width, uint32_t& height,
                      std::string& error);
bool WriteRadianceHdr(const std::string& path, const float* rgb, uint32_t width, uint32_t height, std::string& error);

}  // namespace ptr
