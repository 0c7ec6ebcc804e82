#include "image_writer.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace ptr {
namespace {

struct Rgb {
    float r, g, b;
};

inline float sat(float v) { return std::max(0.0f, std::min(1.0f, v)); }
inline Rgb sat(const Rgb& c) { return {sat(c.r), sat(c.g), sat(c.b)}; }

// Per-channel rational curve helpers written out per operator (ImageWriter.mm:83-137).
Rgb acesFitted(const Rgb& in) {
    // RRT+ODT fit (Hill); matrices are applied row-wise exactly as the reference's applyMatrix
    static const float inM[3][3] = {{0.59719f, 0.07600f, 0.02840f}, {0.35458f, 0.90834f, 0.13383f}, {0.04823f, 0.01566f, 0.83777f}};
    static const float outM[3][3] = {{1.60475f, -0.10208f, -0.00327f}, {-0.53108f, 1.10813f, -0.07276f}, {-0.07367f, -0.00605f, 1.07602f}};
    auto apply = [](const float m[3][3], const Rgb& v) {
        return Rgb{m[0][0] * v.r + m[0][1] * v.g + m[0][2] * v.b, m[1][0] * v.r + m[1][1] * v.g + m[1][2] * v.b,
                   m[2][0] * v.r + m[2][1] * v.g + m[2][2] * v.b};
    };
    auto curve = [](float c) {
        const float a = c * (c + 0.0245786f) - 0.000090537f;
        const float b = c * (0.983729f * c + 0.4329510f) + 0.238081f;
        return a / b;
    };
    Rgb c = apply(inM, in);
    c = {curve(c.r), curve(c.g), curve(c.b)};
    return sat(apply(outM, c));
}

Rgb acesSimple(const Rgb& in) {
    auto curve = [](float c) { return (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f); };
    return sat(Rgb{curve(in.r), curve(in.g), curve(in.b)});
}

Rgb reinhard(const Rgb& c, float whitePoint) {
    const float lum = c.r * 0.2126f + c.g * 0.7152f + c.b * 0.0722f;
    const float denom = 1.0f + lum / std::max(whitePoint, 1e-4f);
    return sat(Rgb{c.r / denom, c.g / denom, c.b / denom});
}

Rgb hable(const Rgb& in) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, W = 11.2f;
    auto curve = [&](float c) { return ((c * (A * c + B)) + C * c + D) / ((c * (A * c + B)) + E * c + F) - D / F; };
    const float white = ((W * (A * W + B)) + C * W + D) / ((W * (A * W + B)) + E * W + F) - D / F;
    return sat(Rgb{curve(in.r) / white, curve(in.g) / white, curve(in.b) / white});
}

Rgb applyTonemap(const Rgb& linear, const TonemapSettings& tm) {
    const float gain = std::pow(2.0f, tm.exposure);
    Rgb c{linear.r * gain, linear.g * gain, linear.b * gain};
    switch (tm.tonemapMode) {
        case 2: c = (tm.acesVariant == 0) ? acesFitted(c) : acesSimple(c); break;
        case 3: c = reinhard(c, tm.reinhardWhitePoint); break;
        case 4: c = hable(c); break;
        default: c = sat(c); break;
    }
    const float gamma = 1.0f / 2.2f;
    c = {std::pow(std::max(c.r, 0.0f), gamma), std::pow(std::max(c.g, 0.0f), gamma), std::pow(std::max(c.b, 0.0f), gamma)};
    return sat(c);
}

inline uint8_t quantize(float v) {
    return static_cast<uint8_t>(std::min(std::max(std::lround(v * 255.0f), 0l), 255l));
}

bool fail(std::string* err, const std::string& msg) {
    if (err) *err = msg;
    return false;
}

bool writePPM(const std::string& path, const float* rgb, uint32_t w, uint32_t h, const TonemapSettings& tm, std::string* err) {
    std::vector<uint8_t> ldr(static_cast<size_t>(w) * h * 3);
    TonemapToLdr(rgb, w * h, tm, ldr.data());
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(err, "Failed to open output file: " + path);
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    fwrite(ldr.data(), 1, ldr.size(), f);
    fclose(f);
    return true;
}

bool writePFM(const std::string& path, const float* rgb, uint32_t w, uint32_t h, std::string* err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(err, "Failed to open output file: " + path);
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);  // negative scale = little-endian
    const size_t rowFloats = static_cast<size_t>(w) * 3;
    for (uint32_t y = h; y-- > 0;) {  // PFM stores the bottom row first
        fwrite(rgb + static_cast<size_t>(y) * rowFloats, sizeof(float), rowFloats, f);
    }
    fclose(f);
    return true;
}

// Byte-stream builder for the EXR header.
struct ByteSink {
    std::vector<uint8_t> bytes;
    void raw(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        bytes.insert(bytes.end(), b, b + n);
    }
    void cstr(const char* s) { raw(s, std::strlen(s) + 1); }
    void u8(uint8_t v) { raw(&v, 1); }
    void u32(uint32_t v) { raw(&v, 4); }
    void i32(int32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
    void attr(const char* name, const char* type, uint32_t size) {
        cstr(name);
        cstr(type);
        u32(size);
    }
};

struct ExrChannel {
    const char* name;
    uint32_t component;  // index into the interleaved source pixel
};

// Uncompressed scanline OpenEXR, FLOAT channels, INCREASING_Y; attribute order follows
// ImageWriter.mm:293-395 so files are byte-identical for identical pixels.
bool writeScanlineExr(const std::string& path, const float* interleaved, uint32_t stride, uint32_t w, uint32_t h,
                      const std::vector<ExrChannel>& channels, const char* colorspace, std::string* err) {
    if (w == 0 || h == 0 || channels.empty() || !interleaved) return fail(err, "Invalid EXR parameters");

    ByteSink hd;
    hd.u32(20000630u);
    hd.u32(2u);
    {
        ByteSink ch;
        for (const ExrChannel& c : channels) {
            ch.cstr(c.name);
            ch.i32(2);  // FLOAT
            ch.u8(0);   // pLinear
            ch.u8(0); ch.u8(0); ch.u8(0);
            ch.i32(1);  // xSampling
            ch.i32(1);  // ySampling
        }
        ch.u8(0);
        hd.attr("channels", "chlist", static_cast<uint32_t>(ch.bytes.size()));
        hd.raw(ch.bytes.data(), ch.bytes.size());
    }
    hd.attr("compression", "compression", 1);
    hd.u8(0);
    for (const char* name : {"dataWindow", "displayWindow"}) {
        hd.attr(name, "box2i", 16);
        hd.i32(0); hd.i32(0);
        hd.i32(static_cast<int32_t>(w) - 1);
        hd.i32(static_cast<int32_t>(h) - 1);
    }
    hd.attr("pixelAspectRatio", "float", 4);
    hd.f32(1.0f);
    hd.attr("screenWindowCenter", "v2f", 8);
    hd.f32(0.0f); hd.f32(0.0f);
    hd.attr("screenWindowWidth", "float", 4);
    hd.f32(1.0f);
    hd.attr("lineOrder", "lineOrder", 1);
    hd.u8(0);
    if (colorspace && colorspace[0] != '\0') {
        const uint32_t n = static_cast<uint32_t>(std::strlen(colorspace) + 1);
        hd.attr("colorspace", "string", n);
        hd.raw(colorspace, n);
    }
    hd.u8(0);  // end of header

    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(err, "Failed to open output file: " + path);
    bool ok = fwrite(hd.bytes.data(), 1, hd.bytes.size(), f) == hd.bytes.size();

    const uint64_t nch = channels.size();
    const uint64_t blockBytes = 8ull + static_cast<uint64_t>(w) * nch * sizeof(float);
    uint64_t offset = hd.bytes.size() + static_cast<uint64_t>(h) * 8ull;
    for (uint32_t y = 0; y < h && ok; ++y) {
        ok = fwrite(&offset, 8, 1, f) == 1;
        offset += blockBytes;
    }
    std::vector<float> line(static_cast<size_t>(w) * nch);
    for (uint32_t y = 0; y < h && ok; ++y) {
        const int32_t yy = static_cast<int32_t>(y);
        const uint32_t packed = static_cast<uint32_t>(line.size() * sizeof(float));
        for (size_t c = 0; c < nch; ++c) {  // channel-planar within a scanline
            const float* src = interleaved + static_cast<size_t>(y) * w * stride + channels[c].component;
            float* dst = line.data() + c * w;
            for (uint32_t x = 0; x < w; ++x) dst[x] = src[static_cast<size_t>(x) * stride];
        }
        ok = fwrite(&yy, 4, 1, f) == 1 && fwrite(&packed, 4, 1, f) == 1 &&
             fwrite(line.data(), sizeof(float), line.size(), f) == line.size();
    }
    fclose(f);
    return ok ? true : fail(err, "Failed writing to EXR file");
}

}  // namespace

void TonemapToLdr(const float* linearRGB, uint32_t pixelCount, const TonemapSettings& tonemap, uint8_t* out) {
    for (uint32_t i = 0; i < pixelCount; ++i) {
        const Rgb c = applyTonemap(Rgb{linearRGB[3 * i], linearRGB[3 * i + 1], linearRGB[3 * i + 2]}, tonemap);
        out[3 * i + 0] = quantize(c.r);
        out[3 * i + 1] = quantize(c.g);
        out[3 * i + 2] = quantize(c.b);
    }
}

bool ParseImageFileFormat(const std::string& value, ImageFileFormat& out) {
    std::string lower;
    for (char c : value) lower.push_back(static_cast<char>(tolower(static_cast<unsigned char>(c))));
    if (lower == "exr") { out = ImageFileFormat::EXR; return true; }
    if (lower == "png") { out = ImageFileFormat::PNG; return true; }
    if (lower == "pfm") { out = ImageFileFormat::PFM; return true; }
    if (lower == "ppm") { out = ImageFileFormat::PPM; return true; }
    return false;
}

const char* FormatExtension(ImageFileFormat format) {
    switch (format) {
        case ImageFileFormat::EXR: return "exr";
        case ImageFileFormat::PNG: return "png";
        case ImageFileFormat::PFM: return "pfm";
        default: return "ppm";
    }
}

bool WriteImage(const std::string& path, ImageFileFormat format, const float* linearRGB, uint32_t width,
                uint32_t height, const TonemapSettings& tonemap, std::string* errorMessage) {
    switch (format) {
        case ImageFileFormat::EXR:
            return writeScanlineExr(path, linearRGB, 3, width, height, {{"B", 2}, {"G", 1}, {"R", 0}}, nullptr, errorMessage);
        case ImageFileFormat::PFM: return writePFM(path, linearRGB, width, height, errorMessage);
        case ImageFileFormat::PPM: return writePPM(path, linearRGB, width, height, tonemap, errorMessage);
        case ImageFileFormat::PNG:
            // The reference encodes PNG through Apple ImageIO (ImageWriter.mm:480-565); SURVEY §8(f) rank 3.
            return fail(errorMessage, "PNG output is not available in this build (use exr, pfm or ppm)");
    }
    return false;
}

bool WriteExrRgba(const std::string& path, const float* rgba, uint32_t width, uint32_t height,
                  const char* colorspace, std::string* errorMessage) {
    return writeScanlineExr(path, rgba, 4, width, height, {{"B", 2}, {"G", 1}, {"R", 0}, {"A", 3}}, colorspace, errorMessage);
}

}  // namespace ptr
