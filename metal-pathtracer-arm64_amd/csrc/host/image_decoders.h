// PNG and baseline-JPEG decoders for glTF material textures (no third-party code, no system libraries).
//
// The reference hands image bytes to the platform (MTKTextureLoader / ImageIO, src/renderer/SceneResources.mm:213-420);
// there is no such service here, so the two formats glTF allows for textures are decoded by hand:
//   PNG  : zlib inflate (stored / fixed / dynamic Huffman blocks), the five scanline filters, colour types 0/2/3/4/6,
//          bit depths 1-16 (16-bit samples keep their high byte), tRNS; Adam7 interlacing is rejected.
//   JPEG : baseline and extended-sequential 8-bit DCT (SOF0 / SOF1), Huffman coding, restart intervals, 1-3 components with
//          sampling factors up to 2x2 (chroma is replicated, not interpolated), JFIF YCbCr -> RGB; progressive / arithmetic /
//          lossless streams are rejected.
// Output is always 8-bit RGBA, row 0 = top.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ptr {

struct DecodedImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;   // width * height * 4
};

bool DecodePng(const uint8_t* data, size_t size, DecodedImage& out, std::string* error = nullptr);
bool DecodeJpeg(const uint8_t* data, size_t size, DecodedImage& out, std::string* error = nullptr);
// Picks the decoder from the magic bytes.
bool DecodeImage(const uint8_t* data, size_t size, DecodedImage& out, std::string* error = nullptr);
// zlib stream -> bytes (exposed for tests).
// maxOutput: a stream that would expand beyond this many bytes is rejected
bool InflateZlib(const uint8_t* data, size_t size, std::vector<uint8_t>& out, std::string* error = nullptr, size_t maxOutput = static_cast<size_t>(1) << 31);

}  // namespace ptr
