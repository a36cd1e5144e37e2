// PNG decoder for the texture front-end (the reference called stbi_load(file, &w, &h, &ch, 3),
// hw8/src/sceneio.cpp:374-379; the stb submodule is absent).  Own inflate (RFC 1951),
// zlib framing (RFC 1950), PNG filters; non-interlaced, colour types 0/2/3/4/6, bit depths 1-16.
// Output is always 3-channel RGB8 with stb's reductions: alpha dropped, gray replicated,
// 16-bit samples truncated to their high byte, sub-byte gray scaled to 0..255.
#include "png.h"
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace rtamd {
namespace {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int cnt = 0;
    uint32_t bits(int n) {
        while (cnt < n) {
            if (p >= end) throw std::runtime_error("PNG: deflate stream truncated");
            buf |= (uint32_t)(*p++) << cnt;
            cnt += 8;
        }
        uint32_t v = buf & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
    void align() { buf = 0; cnt = 0; }
};

struct Huff {
    uint16_t count[16];
    uint16_t symbol[288];
    void build(const uint8_t *lens, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lens[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++)
            if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.bits(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw std::runtime_error("PNG: bad Huffman code");
    }
};

const uint16_t len_base[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint16_t len_extra[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t dist_base[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint16_t dist_extra[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

void inflate_codes(BitReader &br, std::vector<uint8_t> &out, const Huff &lit, const Huff &dist) {
    for (;;) {
        int sym = lit.decode(br);
        if (sym < 256) out.push_back((uint8_t)sym);
        else if (sym == 256) return;
        else {
            sym -= 257;
            if (sym >= 29) throw std::runtime_error("PNG: bad length symbol");
            int len = len_base[sym] + (int)br.bits(len_extra[sym]);
            int ds = dist.decode(br);
            if (ds >= 30) throw std::runtime_error("PNG: bad distance symbol");
            size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
            if (d > out.size()) throw std::runtime_error("PNG: distance too far back");
            size_t from = out.size() - d;
            for (int i = 0; i < len; i++) out.push_back(out[from + i]);
        }
    }
}

std::vector<uint8_t> inflate_zlib(const std::vector<uint8_t> &z, size_t expect) {
    if (z.size() < 6) throw std::runtime_error("PNG: zlib stream too short");
    if ((z[0] & 0x0F) != 8 || ((z[0] << 8 | z[1]) % 31) != 0 || (z[1] & 0x20)) throw std::runtime_error("PNG: bad zlib header");
    BitReader br{z.data() + 2, z.data() + z.size()};
    std::vector<uint8_t> out;
    out.reserve(expect);
    int last;
    do {
        last = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) throw std::runtime_error("PNG: stored block truncated");
            uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xFFFF) != nlen || (size_t)(br.end - br.p) < len) throw std::runtime_error("PNG: bad stored block");
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1) {
            uint8_t lens[320];
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            Huff lit, dist;
            lit.build(lens, 288);
            for (i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
            inflate_codes(br, out, lit, dist);
        } else if (type == 2) {
            int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lens[320];
            memset(lens, 0, sizeof lens);
            for (int i = 0; i < ncode; i++) lens[order[i]] = (uint8_t)br.bits(3);
            Huff cl;
            cl.build(lens, 19);
            uint8_t ll[320];
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = cl.decode(br);
                if (sym < 16) ll[idx++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) throw std::runtime_error("PNG: bad repeat");
                        val = ll[idx - 1];
                        rep = 3 + (int)br.bits(2);
                    } else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (idx + rep > nlen + ndist) throw std::runtime_error("PNG: too many code lengths");
                    while (rep--) ll[idx++] = (uint8_t)val;
                }
            }
            Huff lit, dist;
            lit.build(ll, nlen);
            dist.build(ll + nlen, ndist);
            inflate_codes(br, out, lit, dist);
        } else throw std::runtime_error("PNG: bad block type");
    } while (!last);
    return out;
}

uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

} // namespace

void decode_png(const std::vector<uint8_t> &file, int &width, int &height, std::vector<uint8_t> &rgb) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file");
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4];
        const uint8_t *data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) throw std::runtime_error("PNG: chunk overruns file");
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error("PNG: short IHDR");
            width = (int)be32(data); height = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || width <= 0 || height <= 0) throw std::runtime_error("PNG: missing IHDR");
    if (interlace) throw std::runtime_error("PNG: interlaced images are not supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("PNG: bad colour type");
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        throw std::runtime_error("PNG: unsupported bit depth");
    size_t bpp_bits = (size_t)channels * depth;
    size_t stride = ((size_t)width * bpp_bits + 7) / 8;
    size_t fbpp = (bpp_bits + 7) / 8; // filter byte distance
    std::vector<uint8_t> raw = inflate_zlib(idat, (stride + 1) * height);
    if (raw.size() < (stride + 1) * (size_t)height) throw std::runtime_error("PNG: not enough image data");
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgb.assign((size_t)width * height * 3, 0);
    for (int y = 0; y < height; y++) {
        const uint8_t *in = &raw[(stride + 1) * y];
        int ft = in[0];
        in++;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= fbpp ? cur[i - fbpp] : 0, b = prev[i], c = i >= fbpp ? prev[i - fbpp] : 0, v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: throw std::runtime_error("PNG: bad filter type");
            }
            cur[i] = (uint8_t)v;
        }
        uint8_t *o = &rgb[(size_t)y * width * 3];
        for (int x = 0; x < width; x++) {
            auto sample = [&](int ch) -> int { // 8-bit view of channel ch of pixel x
                if (depth == 8) return cur[(size_t)x * channels + ch];
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2]; // high byte
                int per = 8 / depth, idx = x; // sub-byte: single channel only
                int v = (cur[idx / per] >> (8 - depth * (idx % per + 1))) & ((1 << depth) - 1);
                return v;
            };
            if (ctype == 3) {
                size_t pi = (size_t)sample(0);
                if (pi * 3 + 2 >= plte.size()) throw std::runtime_error("PNG: palette index out of range");
                o[3 * x] = plte[3 * pi]; o[3 * x + 1] = plte[3 * pi + 1]; o[3 * x + 2] = plte[3 * pi + 2];
            } else if (ctype == 0 || ctype == 4) {
                int g = sample(0);
                if (depth < 8) g *= (depth == 1 ? 255 : depth == 2 ? 85 : 17);
                o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = (uint8_t)g;
            } else {
                o[3 * x] = (uint8_t)sample(0); o[3 * x + 1] = (uint8_t)sample(1); o[3 * x + 2] = (uint8_t)sample(2);
            }
        }
        std::swap(prev, cur);
    }
}

std::vector<uint8_t> read_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open file: " + path);
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return data;
}

void load_image_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb) {
    std::vector<uint8_t> file = read_file(path);
    if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) decode_jpeg(file, width, height, rgb);
    else decode_png(file, width, height, rgb);
}

} // namespace rtamd
