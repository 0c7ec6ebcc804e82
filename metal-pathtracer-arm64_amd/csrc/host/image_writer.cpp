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
    uint32_t component;            // index into the interleaved source pixel
    const float* planar = nullptr; // if set: a separate width*height plane instead (ImageWriter.mm "Planar" channels)
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
            float* dst = line.data() + c * w;
            if (channels[c].planar) {
                std::memcpy(dst, channels[c].planar + static_cast<size_t>(y) * w, static_cast<size_t>(w) * sizeof(float));
                continue;
            }
            const float* src = interleaved + static_cast<size_t>(y) * w * stride + channels[c].component;
            for (uint32_t x = 0; x < w; ++x) dst[x] = src[static_cast<size_t>(x) * stride];
        }
        ok = fwrite(&yy, 4, 1, f) == 1 && fwrite(&packed, 4, 1, f) == 1 &&
             fwrite(line.data(), sizeof(float), line.size(), f) == line.size();
    }
    fclose(f);
    return ok ? true : fail(err, "Failed writing to EXR file");
}


// ---------------------------------------------------------------------------------------------------------
// PNG (the reference goes through Apple ImageIO, ImageWriter.mm:480-565: tonemapped 8-bit RGBA, alpha 255, sRGB
// colour space).  Encoder written here: Sub filter per row, zlib stream of fixed-Huffman deflate blocks with a
// hash-chain LZ77 matcher.  Pixel values are the reference's; the byte stream of another encoder can differ.
uint32_t crc32Update(uint32_t crc, const uint8_t* data, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

struct BitSink {
    std::vector<uint8_t> bytes;
    uint32_t acc = 0;
    int fill = 0;
    void put(uint32_t value, int count) {   // LSB-first, as deflate packs everything except Huffman codes
        acc |= value << fill;
        fill += count;
        while (fill >= 8) {
            bytes.push_back(static_cast<uint8_t>(acc & 0xFFu));
            acc >>= 8;
            fill -= 8;
        }
    }
    void putHuffman(uint32_t code, int length) {   // Huffman codes go in MSB-first
        uint32_t reversed = 0;
        for (int i = 0; i < length; ++i) reversed |= ((code >> i) & 1u) << (length - 1 - i);
        put(reversed, length);
    }
    void flush() {
        if (fill > 0) put(0, 8 - fill);
    }
};

void fixedLiteral(BitSink& out, uint32_t symbol) {   // RFC 1951 section 3.2.6
    if (symbol <= 143) out.putHuffman(0x30 + symbol, 8);
    else if (symbol <= 255) out.putHuffman(0x190 + (symbol - 144), 9);
    else if (symbol <= 279) out.putHuffman(symbol - 256, 7);
    else out.putHuffman(0xC0 + (symbol - 280), 8);
}

void fixedMatch(BitSink& out, uint32_t length, uint32_t distance) {
    static const uint16_t lenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int li = 28;
    while (lenBase[li] > length) --li;
    fixedLiteral(out, 257u + static_cast<uint32_t>(li));
    if (lenExtra[li]) out.put(length - lenBase[li], lenExtra[li]);
    int di = 29;
    while (distBase[di] > distance) --di;
    out.putHuffman(static_cast<uint32_t>(di), 5);
    if (distExtra[di]) out.put(distance - distBase[di], distExtra[di]);
}

std::vector<uint8_t> zlibCompress(const std::vector<uint8_t>& in) {
    BitSink out;
    out.put(0x78, 8);   // CMF: deflate, 32 K window
    out.put(0x9C, 8);   // FLG: check bits, default compression
    out.put(1, 1);      // BFINAL
    out.put(1, 2);      // BTYPE = fixed Huffman
    const size_t n = in.size();
    constexpr uint32_t kHashSize = 1u << 15, kWindow = 32768u, kChain = 16u, kMinMatch = 3u, kMaxMatch = 258u;
    std::vector<int64_t> head(kHashSize, -1), prev(n, -1);
    auto hash3 = [&](size_t i) { return ((static_cast<uint32_t>(in[i]) << 10) ^ (static_cast<uint32_t>(in[i + 1]) << 5) ^ in[i + 2]) & (kHashSize - 1); };
    size_t i = 0;
    while (i < n) {
        uint32_t bestLen = 0, bestDist = 0;
        if (i + kMinMatch <= n) {
            const uint32_t h = hash3(i);
            int64_t cand = head[h];
            for (uint32_t chain = 0; cand >= 0 && chain < kChain && i - static_cast<size_t>(cand) <= kWindow; ++chain) {
                const size_t c = static_cast<size_t>(cand);
                uint32_t len = 0;
                const uint32_t limit = static_cast<uint32_t>(std::min<size_t>(kMaxMatch, n - i));
                while (len < limit && in[c + len] == in[i + len]) ++len;
                if (len > bestLen) {
                    bestLen = len;
                    bestDist = static_cast<uint32_t>(i - c);
                    if (len == limit) break;
                }
                cand = prev[c];
            }
            prev[i] = head[h];
            head[h] = static_cast<int64_t>(i);
        }
        if (bestLen >= kMinMatch) {
            fixedMatch(out, bestLen, bestDist);
            for (size_t k = 1; k < bestLen && i + k + kMinMatch <= n; ++k) {   // keep the hash chains current
                const uint32_t h = hash3(i + k);
                prev[i + k] = head[h];
                head[h] = static_cast<int64_t>(i + k);
            }
            i += bestLen;
        } else {
            fixedLiteral(out, in[i]);
            ++i;
        }
    }
    fixedLiteral(out, 256);   // end of block
    out.flush();
    uint32_t a = 1, b = 0;    // Adler-32 of the uncompressed data, big-endian
    for (uint8_t v : in) {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    const uint32_t adler = (b << 16) | a;
    for (int shift = 24; shift >= 0; shift -= 8) out.bytes.push_back(static_cast<uint8_t>((adler >> shift) & 0xFFu));
    return out.bytes;
}

bool writePNG(const std::string& path, const float* rgb, uint32_t w, uint32_t h, const TonemapSettings& tm, std::string* err) {
    // filtered scanlines: filter byte 1 (Sub) + RGBA8
    const size_t rowBytes = static_cast<size_t>(w) * 4;
    std::vector<uint8_t> raw((rowBytes + 1) * h);
    std::vector<uint8_t> row(rowBytes);
    for (uint32_t y = 0; y < h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            const size_t i = static_cast<size_t>(y) * w + x;
            const Rgb c = applyTonemap(Rgb{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, tm);
            row[4 * x + 0] = quantize(c.r);
            row[4 * x + 1] = quantize(c.g);
            row[4 * x + 2] = quantize(c.b);
            row[4 * x + 3] = 255;
        }
        uint8_t* dst = raw.data() + static_cast<size_t>(y) * (rowBytes + 1);
        dst[0] = 1;
        for (size_t k = 0; k < rowBytes; ++k) dst[1 + k] = static_cast<uint8_t>(row[k] - (k >= 4 ? row[k - 4] : 0));
    }
    const std::vector<uint8_t> idat = zlibCompress(raw);

    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(err, "Failed to open output file: " + path);
    bool ok = true;
    auto be32 = [](uint32_t v, uint8_t* p) {
        p[0] = static_cast<uint8_t>(v >> 24), p[1] = static_cast<uint8_t>(v >> 16), p[2] = static_cast<uint8_t>(v >> 8), p[3] = static_cast<uint8_t>(v);
    };
    auto chunk = [&](const char* type, const uint8_t* data, size_t n) {
        uint8_t hdr[8];
        be32(static_cast<uint32_t>(n), hdr);
        std::memcpy(hdr + 4, type, 4);
        uint32_t crc = crc32Update(0xFFFFFFFFu, hdr + 4, 4);
        if (n) crc = crc32Update(crc, data, n);
        uint8_t tail[4];
        be32(crc ^ 0xFFFFFFFFu, tail);
        ok = ok && fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    ok = fwrite(signature, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(w, ihdr);
    be32(h, ihdr + 4);
    ihdr[8] = 8;    // bit depth
    ihdr[9] = 6;    // colour type RGBA
    ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    const uint8_t intent = 0;   // sRGB chunk, perceptual (kCGRenderingIntentDefault)
    chunk("sRGB", &intent, 1);
    chunk("IDAT", idat.data(), idat.size());
    chunk("IEND", nullptr, 0);
    fclose(f);
    return ok ? true : fail(err, "Failed to finalize PNG file");
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
        case ImageFileFormat::PNG: return writePNG(path, linearRGB, width, height, tonemap, errorMessage);
    }
    return false;
}

bool WriteExrRgba(const std::string& path, const float* rgba, uint32_t width, uint32_t height,
                  const char* colorspace, std::string* errorMessage) {
    return writeScanlineExr(path, rgba, 4, width, height, {{"B", 2}, {"G", 1}, {"R", 0}, {"A", 3}}, colorspace, errorMessage);
}

bool WriteExrMultilayer(const std::string& path, const float* rgba, uint32_t width, uint32_t height, const float* sampleCount,
                        const char* colorspace, std::string* errorMessage) {
    if (!sampleCount) return WriteExrRgba(path, rgba, width, height, colorspace, errorMessage);
    return writeScanlineExr(path, rgba, 4, width, height, {{"B", 2}, {"G", 1}, {"R", 0}, {"A", 3}, {"SAMPLES", 0, sampleCount}}, colorspace,
                            errorMessage);
}


bool WriteExrAovs(const std::string& path, const float* rgb, const float* albedoRgba, const float* normalRgba, uint32_t width, uint32_t height,
                  std::string* errorMessage) {
    if (!rgb || !albedoRgba || !normalRgba || width == 0 || height == 0) return fail(errorMessage, "Invalid EXR parameters");
    const size_t n = static_cast<size_t>(width) * height;
    // planes: beauty R G B, albedo R G B, normal X Y Z (decoded from the 0.5 + 0.5 n encoding), depth
    std::vector<float> planes(n * 10u);
    float* p[10];
    for (int k = 0; k < 10; ++k) p[k] = planes.data() + static_cast<size_t>(k) * n;
    for (size_t i = 0; i < n; ++i) {
        p[0][i] = rgb[i * 3 + 0];
        p[1][i] = rgb[i * 3 + 1];
        p[2][i] = rgb[i * 3 + 2];
        p[3][i] = albedoRgba[i * 4 + 0];
        p[4][i] = albedoRgba[i * 4 + 1];
        p[5][i] = albedoRgba[i * 4 + 2];
        const bool hit = albedoRgba[i * 4 + 3] > 0.5f;
        p[6][i] = hit ? normalRgba[i * 4 + 0] * 2.0f - 1.0f : 0.0f;
        p[7][i] = hit ? normalRgba[i * 4 + 1] * 2.0f - 1.0f : 0.0f;
        p[8][i] = hit ? normalRgba[i * 4 + 2] * 2.0f - 1.0f : 0.0f;
        p[9][i] = normalRgba[i * 4 + 3];
    }
    // channel list in the alphabetical order the OpenEXR format asks for
    return writeScanlineExr(path, rgb, 3, width, height,
                            {{"B", 0, p[2]}, {"G", 0, p[1]}, {"R", 0, p[0]}, {"albedo.B", 0, p[5]}, {"albedo.G", 0, p[4]}, {"albedo.R", 0, p[3]},
                             {"depth.Z", 0, p[9]}, {"normal.X", 0, p[6]}, {"normal.Y", 0, p[7]}, {"normal.Z", 0, p[8]}},
                            "Linear sRGB", errorMessage);
}

}  // namespace ptr
