#include "image_decoders.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ptr {
namespace {

bool fail(std::string* err, const char* msg) {
    if (err) *err = msg;
    return false;
}

// ------------------------------------------------------------------------------------------------ inflate (RFC 1950 / 1951)
struct BitReader {
    const uint8_t* p;
    size_t n, at = 0;
    uint32_t acc = 0;
    int bits = 0;
    bool bad = false;
    uint32_t get(int count) {   // LSB first
        while (bits < count) {
            if (at >= n) {
                bad = true;
                return 0;
            }
            acc |= static_cast<uint32_t>(p[at++]) << bits;
            bits += 8;
        }
        const uint32_t v = acc & ((count == 32) ? 0xFFFFFFFFu : ((1u << count) - 1u));
        acc >>= count;
        bits -= count;
        return v;
    }
    void alignByte() {
        acc = 0;
        bits = 0;
    }
};

// canonical Huffman decoder, bit-by-bit over (count, symbol) tables: small and fast enough for texture-sized inputs
struct Huffman {
    uint16_t count[16] = {0};
    uint16_t symbol[288] = {0};
    bool build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) ++count[lengths[i]];
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;   // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
        for (int i = 0; i < n; ++i) {
            if (lengths[i]) symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
        }
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= static_cast<int>(br.get(1));
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflateBlockData(BitReader& br, const Huffman& lit, const Huffman& dist, std::vector<uint8_t>& out, size_t cap) {
    while (true) {
        const int sym = lit.decode(br);
        if (sym < 0) return false;
        if (sym < 256) {
            if (out.size() >= cap) return false;
            out.push_back(static_cast<uint8_t>(sym));
        } else if (sym == 256) {
            return true;
        } else {
            const int li = sym - 257;
            if (li >= 29) return false;
            const uint32_t length = kLenBase[li] + br.get(kLenExtra[li]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds >= 30) return false;
            const uint32_t distance = kDistBase[ds] + br.get(kDistExtra[ds]);
            if (br.bad || distance > out.size() || out.size() + length > cap) return false;
            size_t from = out.size() - distance;
            for (uint32_t i = 0; i < length; ++i) out.push_back(out[from++]);
        }
    }
}

// cap: the most output the caller can use; a stream that expands beyond it is rejected (a few KB of deflate can expand ~1000:1)
bool inflateRaw(BitReader& br, std::vector<uint8_t>& out, size_t cap) {
    bool last = false;
    while (!last) {
        last = br.get(1) != 0;
        const uint32_t type = br.get(2);
        if (br.bad) return false;
        if (type == 0) {
            br.alignByte();
            if (br.at + 4 > br.n) return false;
            const uint32_t len = br.p[br.at] | (br.p[br.at + 1] << 8), nlen = br.p[br.at + 2] | (br.p[br.at + 3] << 8);
            br.at += 4;
            if ((len ^ 0xFFFFu) != nlen || br.at + len > br.n || out.size() + len > cap) return false;
            out.insert(out.end(), br.p + br.at, br.p + br.at + len);
            br.at += len;
        } else if (type == 1) {
            uint8_t lengths[288];
            for (int i = 0; i < 144; ++i) lengths[i] = 8;
            for (int i = 144; i < 256; ++i) lengths[i] = 9;
            for (int i = 256; i < 280; ++i) lengths[i] = 7;
            for (int i = 280; i < 288; ++i) lengths[i] = 8;
            Huffman lit, dist;
            lit.build(lengths, 288);
            uint8_t dl[30];
            for (int i = 0; i < 30; ++i) dl[i] = 5;
            dist.build(dl, 30);
            if (!inflateBlockData(br, lit, dist, out, cap)) return false;
        } else if (type == 2) {
            const int nlen = static_cast<int>(br.get(5)) + 257, ndist = static_cast<int>(br.get(5)) + 1, ncode = static_cast<int>(br.get(4)) + 4;
            if (br.bad || nlen > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320] = {0};
            for (int i = 0; i < ncode; ++i) lengths[order[i]] = static_cast<uint8_t>(br.get(3));
            Huffman lencode;
            if (!lencode.build(lengths, 19)) return false;
            int index = 0;
            uint8_t ll[320] = {0};
            while (index < nlen + ndist) {
                const int sym = lencode.decode(br);
                if (sym < 0) return false;
                if (sym < 16) {
                    ll[index++] = static_cast<uint8_t>(sym);
                } else {
                    int prev = 0, rep = 0;
                    if (sym == 16) {
                        if (index == 0) return false;
                        prev = ll[index - 1];
                        rep = 3 + static_cast<int>(br.get(2));
                    } else if (sym == 17) {
                        rep = 3 + static_cast<int>(br.get(3));
                    } else {
                        rep = 11 + static_cast<int>(br.get(7));
                    }
                    if (br.bad || index + rep > nlen + ndist) return false;
                    while (rep--) ll[index++] = static_cast<uint8_t>(prev);
                }
            }
            if (ll[256] == 0) return false;
            Huffman lit, dist;
            if (!lit.build(ll, nlen)) return false;
            dist.build(ll + nlen, ndist);   // an incomplete distance code is legal (a single distance)
            if (!inflateBlockData(br, lit, dist, out, cap)) return false;
        } else {
            return false;
        }
        if (br.bad) return false;
    }
    return true;
}

uint32_t be32(const uint8_t* p) { return (static_cast<uint32_t>(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

}  // namespace

bool InflateZlib(const uint8_t* data, size_t size, std::vector<uint8_t>& out, std::string* error, size_t maxOutput) {
    out.clear();
    if (size < 6) return fail(error, "zlib stream too short");
    if ((data[0] & 0x0F) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) return fail(error, "bad zlib header");
    BitReader br{data + 2, size - 2};
    if (!inflateRaw(br, out, maxOutput)) return fail(error, "corrupt deflate stream (or output beyond the expected size)");
    // adler32 of the output follows (after alignment); verify when present
    br.alignByte();
    if (br.at + 4 <= br.n) {
        uint32_t a = 1, b = 0;
        for (uint8_t c : out) {
            a = (a + c) % 65521u;
            b = (b + a) % 65521u;
        }
        if (be32(br.p + br.at) != ((b << 16) | a)) return fail(error, "zlib checksum mismatch");
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ PNG
bool DecodePng(const uint8_t* data, size_t size, DecodedImage& out, std::string* error) {
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || std::memcmp(data, magic, 8) != 0) return fail(error, "not a PNG file");
    uint32_t width = 0, height = 0;
    int depth = 0, colorType = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool haveHeader = false;
    size_t at = 8;
    while (at + 12 <= size) {
        const uint32_t len = be32(data + at);
        const uint8_t* type = data + at + 4;
        const uint8_t* body = data + at + 8;
        if (at + 12 + static_cast<size_t>(len) > size) return fail(error, "truncated PNG chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return fail(error, "bad IHDR");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colorType = body[9];
            interlace = body[12];
            haveHeader = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + static_cast<size_t>(len);
    }
    if (!haveHeader || width == 0 || height == 0 || width > 16384 || height > 16384) return fail(error, "bad PNG dimensions");
    if (interlace != 0) return fail(error, "interlaced PNG is not supported");
    int channels;
    switch (colorType) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return fail(error, "bad PNG colour type");
    }
    if (!(depth == 8 || depth == 16 || ((colorType == 0 || colorType == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        return fail(error, "unsupported PNG bit depth");
    }
    if (colorType == 3 && (depth == 16 || palette.size() < 3)) return fail(error, "bad PNG palette");
    const size_t bpp = std::max<size_t>(1, static_cast<size_t>(channels) * depth / 8);             // filter unit
    const size_t rowBytes = (static_cast<size_t>(width) * channels * depth + 7) / 8;
    std::vector<uint8_t> raw;
    if (!InflateZlib(idat.data(), idat.size(), raw, error, (rowBytes + 1) * height)) return false;
    if (raw.size() < (rowBytes + 1) * height) return fail(error, "PNG pixel data too short");

    // undo the scanline filters in place
    std::vector<uint8_t> prevRow(rowBytes, 0);
    out.width = width;
    out.height = height;
    out.rgba.assign(static_cast<size_t>(width) * height * 4u, 255);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* row = raw.data() + static_cast<size_t>(y) * (rowBytes + 1);
        const int filter = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < rowBytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prevRow[i], c = i >= bpp ? prevRow[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return fail(error, "bad PNG filter type");
            }
            cur[i] = static_cast<uint8_t>(cur[i] + pred);
        }
        std::memcpy(prevRow.data(), cur, rowBytes);
        uint8_t* dst = out.rgba.data() + static_cast<size_t>(y) * width * 4u;
        auto sample = [&](uint32_t x, int ch) -> uint32_t {   // 8-bit value of channel ch of pixel x
            if (depth == 8) return cur[static_cast<size_t>(x) * channels + ch];
            if (depth == 16) return cur[(static_cast<size_t>(x) * channels + ch) * 2];   // high byte
            const size_t bit = static_cast<size_t>(x) * depth;
            const uint32_t v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
            return colorType == 3 ? v : v * 255u / ((1u << depth) - 1u);
        };
        for (uint32_t x = 0; x < width; ++x) {
            uint8_t* px = dst + static_cast<size_t>(x) * 4u;
            switch (colorType) {
                case 0: {
                    const uint32_t g = sample(x, 0);
                    px[0] = px[1] = px[2] = static_cast<uint8_t>(g);
                    if (trns.size() >= 2 && depth <= 8 && static_cast<uint32_t>((trns[0] << 8) | trns[1]) == (depth == 8 ? g : g * ((1u << depth) - 1u) / 255u)) px[3] = 0;
                    break;
                }
                case 2:
                    px[0] = static_cast<uint8_t>(sample(x, 0));
                    px[1] = static_cast<uint8_t>(sample(x, 1));
                    px[2] = static_cast<uint8_t>(sample(x, 2));
                    if (trns.size() >= 6 && depth == 8 && trns[1] == px[0] && trns[3] == px[1] && trns[5] == px[2]) px[3] = 0;
                    break;
                case 3: {
                    const uint32_t idx = sample(x, 0);
                    if (static_cast<size_t>(idx) * 3 + 2 < palette.size()) {
                        px[0] = palette[idx * 3];
                        px[1] = palette[idx * 3 + 1];
                        px[2] = palette[idx * 3 + 2];
                    } else {
                        px[0] = px[1] = px[2] = 0;
                    }
                    if (idx < trns.size()) px[3] = trns[idx];
                    break;
                }
                case 4: {
                    const uint32_t g = sample(x, 0);
                    px[0] = px[1] = px[2] = static_cast<uint8_t>(g);
                    px[3] = static_cast<uint8_t>(sample(x, 1));
                    break;
                }
                default:
                    px[0] = static_cast<uint8_t>(sample(x, 0));
                    px[1] = static_cast<uint8_t>(sample(x, 1));
                    px[2] = static_cast<uint8_t>(sample(x, 2));
                    px[3] = static_cast<uint8_t>(sample(x, 3));
                    break;
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ JPEG (baseline / extended sequential)
namespace {

struct JpegHuff {
    // ITU T.81 Annex F decoding tables
    int mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    bool present = false;
    void build(const uint8_t* counts, const uint8_t* vals, int total) {
        std::memcpy(values, vals, static_cast<size_t>(total));
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct JpegBits {
    const uint8_t* p;
    size_t n, at;
    uint32_t acc = 0;
    int bits = 0;
    bool hitMarker = false;
    int get1() {
        if (bits == 0) {
            uint8_t b = 0;
            if (at < n && !hitMarker) {
                b = p[at++];
                if (b == 0xFF) {
                    const uint8_t next = at < n ? p[at] : 0;
                    if (next == 0x00) {
                        ++at;   // stuffed zero
                    } else {
                        hitMarker = true;   // a marker: feed zeros from here on (the caller handles RSTn)
                        --at;
                        b = 0;
                    }
                }
            }
            acc = b;
            bits = 8;
        }
        --bits;
        return (acc >> bits) & 1;
    }
    int get(int count) {
        int v = 0;
        while (count--) v = (v << 1) | get1();
        return v;
    }
    void reset() {
        bits = 0;
        hitMarker = false;
    }
};

int jpegDecodeSymbol(JpegBits& br, const JpegHuff& h) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.get1();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}

int jpegExtend(int v, int t) { return (t > 0 && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// separable float inverse DCT of one dequantised block (natural order in, samples 0..255 out)
void idct8x8(const float* in, uint8_t* out, size_t stride) {
    static float basis[8][8];
    static bool ready = false;
    if (!ready) {
        for (int x = 0; x < 8; ++x) {
            for (int u = 0; u < 8; ++u) basis[x][u] = (u == 0 ? 0.35355339059327373f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        }
        ready = true;
    }
    float tmp[64];
    for (int v = 0; v < 8; ++v) {       // rows: along u
        for (int x = 0; x < 8; ++x) {
            float s = 0.0f;
            for (int u = 0; u < 8; ++u) s += basis[x][u] * in[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    }
    for (int x = 0; x < 8; ++x) {       // columns: along v
        for (int y = 0; y < 8; ++y) {
            float s = 0.0f;
            for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[v * 8 + x];
            const int q = static_cast<int>(std::floor(s + 128.5f));
            out[static_cast<size_t>(y) * stride + x] = static_cast<uint8_t>(std::min(std::max(q, 0), 255));
        }
    }
}

}  // namespace

bool DecodeJpeg(const uint8_t* data, size_t size, DecodedImage& out, std::string* error) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail(error, "not a JPEG file");
    uint16_t quant[4][64] = {{0}};
    JpegHuff dc[4], ac[4];
    struct Component {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
        int pred = 0;
        uint32_t planeW = 0, planeH = 0;
        std::vector<uint8_t> plane;
    } comp[3];
    int ncomp = 0;
    uint32_t width = 0, height = 0;
    int restartInterval = 0;
    bool haveFrame = false;
    size_t at = 2;
    while (at + 4 <= size) {
        if (data[at] != 0xFF) return fail(error, "corrupt JPEG (marker expected)");
        while (at < size && data[at] == 0xFF) ++at;
        if (at >= size) break;
        const int marker = data[at++];
        if (marker == 0xD9) break;
        if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
        if (at + 2 > size) return fail(error, "truncated JPEG");
        const size_t len = (static_cast<size_t>(data[at]) << 8) | data[at + 1];
        if (len < 2 || at + len > size) return fail(error, "truncated JPEG segment");
        const uint8_t* seg = data + at + 2;
        const size_t segLen = len - 2;
        if (marker == 0xDB) {
            size_t k = 0;
            while (k < segLen) {
                const int pq = seg[k] >> 4, tq = seg[k] & 15;
                ++k;
                if (tq > 3 || k + (pq ? 128u : 64u) > segLen) return fail(error, "bad JPEG quantisation table");
                for (int i = 0; i < 64; ++i) {
                    quant[tq][i] = pq ? static_cast<uint16_t>((seg[k] << 8) | seg[k + 1]) : seg[k];
                    k += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC4) {
            size_t k = 0;
            while (k + 17 <= segLen) {
                const int tc = seg[k] >> 4, th = seg[k] & 15;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += seg[k + 1 + i];
                if (th > 3 || tc > 1 || total > 256 || k + 17 + static_cast<size_t>(total) > segLen) return fail(error, "bad JPEG Huffman table");
                (tc ? ac[th] : dc[th]).build(seg + k + 1, seg + k + 17, total);
                k += 17 + static_cast<size_t>(total);
            }
        } else if (marker == 0xC0 || marker == 0xC1) {
            if (segLen < 6 || seg[0] != 8) return fail(error, "unsupported JPEG sample precision");
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || segLen < 6 + static_cast<size_t>(ncomp) * 3) return fail(error, "unsupported JPEG component count");
            for (int i = 0; i < ncomp; ++i) {
                comp[i].id = seg[6 + i * 3];
                comp[i].h = seg[7 + i * 3] >> 4;
                comp[i].v = seg[7 + i * 3] & 15;
                comp[i].tq = seg[8 + i * 3] & 3;
                if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2) return fail(error, "unsupported JPEG sampling factors");
            }
            haveFrame = true;
        } else if (marker == 0xC2 || (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC)) {
            return fail(error, "progressive / lossless / arithmetic JPEG is not supported");
        } else if (marker == 0xDD) {
            if (segLen >= 2) restartInterval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {
            if (!haveFrame || width == 0 || height == 0 || width > 16384 || height > 16384) return fail(error, "JPEG scan before frame header");
            if (segLen < 1) return fail(error, "truncated JPEG scan header");
            const int ns = seg[0];
            if (ns != ncomp || segLen < 1 + static_cast<size_t>(ns) * 2 + 3) return fail(error, "unsupported JPEG scan layout");
            for (int i = 0; i < ns; ++i) {
                const int id = seg[1 + i * 2];
                for (int c = 0; c < ncomp; ++c) {
                    if (comp[c].id == id) {
                        comp[c].td = seg[2 + i * 2] >> 4;
                        comp[c].ta = seg[2 + i * 2] & 15;
                    }
                }
            }
            int hmax = 1, vmax = 1;
            for (int c = 0; c < ncomp; ++c) {
                hmax = std::max(hmax, comp[c].h);
                vmax = std::max(vmax, comp[c].v);
                if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present) return fail(error, "JPEG scan refers to a missing table");
            }
            const uint32_t mcuW = 8u * hmax, mcuH = 8u * vmax;
            const uint32_t mcusX = (width + mcuW - 1) / mcuW, mcusY = (height + mcuH - 1) / mcuH;
            for (int c = 0; c < ncomp; ++c) {
                comp[c].planeW = mcusX * 8u * comp[c].h;
                comp[c].planeH = mcusY * 8u * comp[c].v;
                comp[c].plane.assign(static_cast<size_t>(comp[c].planeW) * comp[c].planeH, 128);
                comp[c].pred = 0;
            }
            JpegBits br{data, size, at + len};
            int untilRestart = restartInterval;
            for (uint32_t my = 0; my < mcusY; ++my) {
                for (uint32_t mx = 0; mx < mcusX; ++mx) {
                    if (restartInterval && untilRestart == 0) {
                        // expect RSTn: skip to it, reset predictors
                        br.reset();
                        while (br.at + 1 < size && !(data[br.at] == 0xFF && data[br.at + 1] >= 0xD0 && data[br.at + 1] <= 0xD7)) ++br.at;
                        if (br.at + 1 < size) br.at += 2;
                        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                        untilRestart = restartInterval;
                    }
                    for (int c = 0; c < ncomp; ++c) {
                        for (int by = 0; by < comp[c].v; ++by) {
                            for (int bx = 0; bx < comp[c].h; ++bx) {
                                float block[64] = {0.0f};
                                const int t = jpegDecodeSymbol(br, dc[comp[c].td]);
                                if (t < 0 || t > 11) return fail(error, "corrupt JPEG entropy data");
                                const int diff = t ? jpegExtend(br.get(t), t) : 0;
                                comp[c].pred += diff;
                                block[0] = static_cast<float>(comp[c].pred * quant[comp[c].tq][0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = jpegDecodeSymbol(br, ac[comp[c].ta]);
                                    if (rs < 0) return fail(error, "corrupt JPEG entropy data");
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) {
                                        if (r == 15) {
                                            k += 16;
                                            continue;
                                        }
                                        break;
                                    }
                                    k += r;
                                    if (k > 63) return fail(error, "corrupt JPEG entropy data");
                                    block[kZigzag[k]] = static_cast<float>(jpegExtend(br.get(s), s) * quant[comp[c].tq][k]);
                                    ++k;
                                }
                                const uint32_t px = (mx * comp[c].h + bx) * 8u, py = (my * comp[c].v + by) * 8u;
                                idct8x8(block, comp[c].plane.data() + static_cast<size_t>(py) * comp[c].planeW + px, comp[c].planeW);
                            }
                        }
                    }
                    if (restartInterval) --untilRestart;
                }
            }
            // colour conversion (JFIF YCbCr, full range), chroma replicated
            out.width = width;
            out.height = height;
            out.rgba.assign(static_cast<size_t>(width) * height * 4u, 255);
            for (uint32_t y = 0; y < height; ++y) {
                for (uint32_t x = 0; x < width; ++x) {
                    uint8_t* px = out.rgba.data() + (static_cast<size_t>(y) * width + x) * 4u;
                    auto at2 = [&](int c) {
                        const uint32_t sx = x * comp[c].h / hmax, sy = y * comp[c].v / vmax;
                        return static_cast<float>(comp[c].plane[static_cast<size_t>(sy) * comp[c].planeW + sx]);
                    };
                    if (ncomp == 1) {
                        px[0] = px[1] = px[2] = static_cast<uint8_t>(at2(0));
                    } else {
                        const float Y = at2(0), cb = at2(1) - 128.0f, cr = at2(2) - 128.0f;
                        auto clamp8 = [](float v) { return static_cast<uint8_t>(std::min(std::max(static_cast<int>(std::floor(v + 0.5f)), 0), 255)); };
                        px[0] = clamp8(Y + 1.402f * cr);
                        px[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
                        px[2] = clamp8(Y + 1.772f * cb);
                    }
                }
            }
            return true;
        }
        at += len;
    }
    return fail(error, "JPEG has no scan");
}

bool DecodeImage(const uint8_t* data, size_t size, DecodedImage& out, std::string* error) {
    if (size >= 8 && data[0] == 0x89 && data[1] == 'P') return DecodePng(data, size, out, error);
    if (size >= 3 && data[0] == 0xFF && data[1] == 0xD8) return DecodeJpeg(data, size, out, error);
    return fail(error, "unknown image format (PNG and JPEG are supported)");
}

}  // namespace ptr
