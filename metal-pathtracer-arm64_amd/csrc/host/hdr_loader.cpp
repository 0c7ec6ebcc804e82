#include "hdr_loader.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace ptr {
namespace {

struct FileCloser {
    FILE* f;
    ~FileCloser() {
        if (f) fclose(f);
    }
};

bool readLine(FILE* f, std::string& out) {
    out.clear();
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '\n') return true;
        if (c != '\r') out.push_back(static_cast<char>(c));
    }
    return !out.empty();
}

inline void rgbeToFloat(const uint8_t px[4], float* dst) {
    if (px[3] == 0) {
        dst[0] = dst[1] = dst[2] = 0.0f;
    } else {
        // Ward's rgbe.c convention: mantissa * 2^(e - 136)
        const float f = std::ldexp(1.0f, static_cast<int>(px[3]) - (128 + 8));
        dst[0] = px[0] * f;
        dst[1] = px[1] * f;
        dst[2] = px[2] * f;
    }
    dst[3] = 1.0f;
}

bool loadRadiance(FILE* f, std::vector<float>& rgba, uint32_t& width, uint32_t& height, std::string& error) {
    std::string line;
    if (!readLine(f, line) || (line.rfind("#?", 0) != 0)) {
        error = "not a Radiance HDR file";
        return false;
    }
    bool formatOk = false;
    while (readLine(f, line)) {
        if (line.empty()) break;
        if (line.rfind("FORMAT=", 0) == 0) {
            formatOk = (line == "FORMAT=32-bit_rle_rgbe");
        }
    }
    if (!formatOk) {
        error = "unsupported Radiance FORMAT (need 32-bit_rle_rgbe)";
        return false;
    }
    if (!readLine(f, line)) {
        error = "missing resolution line";
        return false;
    }
    int h = 0, w = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) {
        error = "unsupported resolution line: " + line;
        return false;
    }
    width = static_cast<uint32_t>(w);
    height = static_cast<uint32_t>(h);
    rgba.assign(static_cast<size_t>(w) * h * 4, 0.0f);
    std::vector<uint8_t> scan(static_cast<size_t>(w) * 4);

    for (int y = 0; y < h; ++y) {
        uint8_t head[4];
        if (fread(head, 1, 4, f) != 4) {
            error = "truncated HDR data";
            return false;
        }
        const bool rle = (w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && (head[2] & 0x80) == 0);
        if (rle) {
            if (((head[2] << 8) | head[3]) != w) {
                error = "HDR scanline width mismatch";
                return false;
            }
            for (int c = 0; c < 4; ++c) {
                int x = 0;
                while (x < w) {
                    int count = fgetc(f);
                    if (count == EOF) {
                        error = "truncated HDR RLE data";
                        return false;
                    }
                    if (count > 128) {
                        count -= 128;
                        const int value = fgetc(f);
                        if (value == EOF || x + count > w) {
                            error = "corrupt HDR RLE run";
                            return false;
                        }
                        for (int i = 0; i < count; ++i) scan[static_cast<size_t>(x++) * 4 + c] = static_cast<uint8_t>(value);
                    } else {
                        if (count == 0 || x + count > w) {
                            error = "corrupt HDR RLE literal";
                            return false;
                        }
                        for (int i = 0; i < count; ++i) {
                            const int value = fgetc(f);
                            if (value == EOF) {
                                error = "truncated HDR RLE data";
                                return false;
                            }
                            scan[static_cast<size_t>(x++) * 4 + c] = static_cast<uint8_t>(value);
                        }
                    }
                }
            }
        } else {
            std::memcpy(scan.data(), head, 4);
            if (w > 1 && fread(scan.data() + 4, 4, static_cast<size_t>(w) - 1, f) != static_cast<size_t>(w) - 1) {
                error = "truncated flat HDR data";
                return false;
            }
        }
        float* row = rgba.data() + static_cast<size_t>(y) * w * 4;
        for (int x = 0; x < w; ++x) {
            rgbeToFloat(&scan[static_cast<size_t>(x) * 4], row + static_cast<size_t>(x) * 4);
        }
    }
    return true;
}

bool loadPfm(FILE* f, std::vector<float>& rgba, uint32_t& width, uint32_t& height, std::string& error) {
    std::string magic, dims, scaleLine;
    if (!readLine(f, magic) || !readLine(f, dims) || !readLine(f, scaleLine)) {
        error = "truncated PFM header";
        return false;
    }
    const int channels = (magic == "PF") ? 3 : (magic == "Pf" ? 1 : 0);
    int w = 0, h = 0;
    if (channels == 0 || sscanf(dims.c_str(), "%d %d", &w, &h) != 2 || w <= 0 || h <= 0) {
        error = "bad PFM header";
        return false;
    }
    const float scale = std::strtof(scaleLine.c_str(), nullptr);
    if (!(scale < 0.0f)) {
        error = "big-endian PFM not supported";
        return false;
    }
    width = static_cast<uint32_t>(w);
    height = static_cast<uint32_t>(h);
    rgba.assign(static_cast<size_t>(w) * h * 4, 1.0f);
    std::vector<float> row(static_cast<size_t>(w) * channels);
    for (int y = h - 1; y >= 0; --y) {  // PFM rows are bottom-to-top
        if (fread(row.data(), sizeof(float), row.size(), f) != row.size()) {
            error = "truncated PFM data";
            return false;
        }
        float* dst = rgba.data() + static_cast<size_t>(y) * w * 4;
        for (int x = 0; x < w; ++x) {
            for (int c = 0; c < 3; ++c) {
                dst[static_cast<size_t>(x) * 4 + c] = row[static_cast<size_t>(x) * channels + (channels == 3 ? c : 0)];
            }
        }
    }
    return true;
}

}  // namespace

bool LoadHdrImageRgba(const std::string& path, std::vector<float>& rgba, uint32_t& width, uint32_t& height,
                      std::string& error) {
    FileCloser file{fopen(path.c_str(), "rb")};
    if (!file.f) {
        error = "cannot open environment map: " + path;
        return false;
    }
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    for (char& c : ext) c = static_cast<char>(tolower(static_cast<unsigned char>(c)));
    if (ext == ".pfm") {
        return loadPfm(file.f, rgba, width, height, error);
    }
    if (ext == ".hdr" || ext == ".pic") {
        return loadRadiance(file.f, rgba, width, height, error);
    }
    error = "unsupported environment map format (need .hdr or .pfm): " + path;
    return false;
}

bool WriteRadianceHdr(const std::string& path, const float* rgb, uint32_t width, uint32_t height, std::string& error) {
    FileCloser file{fopen(path.c_str(), "wb")};
    if (!file.f) {
        error = "cannot open for writing: " + path;
        return false;
    }
    fprintf(file.f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", height, width);
    std::vector<uint8_t> row(static_cast<size_t>(width) * 4);
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const float* p = rgb + (static_cast<size_t>(y) * width + x) * 3;
            const float m = std::fmax(p[0], std::fmax(p[1], p[2]));
            uint8_t* o = &row[static_cast<size_t>(x) * 4];
            if (m < 1e-32f) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {
                int e = 0;
                const float s = std::frexp(m, &e) * 256.0f / m;
                o[0] = static_cast<uint8_t>(p[0] * s);
                o[1] = static_cast<uint8_t>(p[1] * s);
                o[2] = static_cast<uint8_t>(p[2] * s);
                o[3] = static_cast<uint8_t>(e + 128);
            }
        }
        fwrite(row.data(), 1, row.size(), file.f);  // flat (non-RLE) scanlines
    }
    return true;
}

}  // namespace ptr
