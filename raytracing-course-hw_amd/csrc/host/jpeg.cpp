// Baseline / extended-sequential JPEG (8-bit, Huffman) -> 3-channel RGB8, for glTF textures.
//
// The reference decodes textures with stb_image (stbi_load(..., 3), hw8/src/sceneio.cpp:374-379), which is not vendored
// here and cannot be run, so this decoder follows stb_image's published JPEG pipeline step by step — its integer IDCT
// (12-bit constants, two passes, +512 >> 10 then +65536+(128<<17) >> 17), its triangle-filter chroma upsampling
// ((3*near + far + 2) >> 2 horizontally / vertically, (3*t0 + t1 + 8) >> 4 for 2x2) and its 20-bit fixed-point YCbCr
// conversion — so that texel bytes agree with what the reference would load.  PARITY UNPINNED: no JPEG asset ships with
// the reference and stb_image cannot be built here; the test checks this decoder against libjpeg (via PIL) to a small
// tolerance only.  Progressive JPEGs, CMYK/Adobe files and 12-bit precision are rejected with an error.
#include "png.h"
#include <cstring>
#include <stdexcept>

namespace rtamd {
namespace {

struct Huff {
    uint8_t size[257];
    uint16_t code[256];
    uint8_t values[256];
    int maxcode[18];
    int delta[17];
    int count = 0;
    bool present = false;
    void build(const int *counts, const uint8_t *vals) { // generate codes in canonical order (JPEG Annex C)
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < counts[i]; j++) size[k++] = (uint8_t)(i + 1);
        size[k] = 0;
        count = k;
        memcpy(values, vals, (size_t)k);
        int code = 0;
        k = 0;
        for (int j = 1; j <= 16; j++) {
            delta[j] = k - code;
            if (size[k] == j) {
                while (size[k] == j) code++, this->code[k] = (uint16_t)(code - 1), k++;
                if (code - 1 >= (1 << j)) throw std::runtime_error("JPEG: bad Huffman code lengths");
            }
            maxcode[j] = code << (16 - j);
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int w2 = 0, h2 = 0; // padded plane size (multiples of the MCU footprint)
    std::vector<uint8_t> data;
};

struct Decoder {
    const uint8_t *p, *end;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    uint8_t marker = 0;
    uint16_t dequant[4][64];
    Huff dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart_interval = 0;

    Decoder(const uint8_t *b, const uint8_t *e) : p(b), end(e) { memset(dequant, 0, sizeof dequant); }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { int a = get8(); return (a << 8) | get8(); }

    void grow_bits() {
        while (bitcnt <= 24) {
            int b = hit_marker ? 0 : get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = (uint8_t)c; hit_marker = true; b = 0; }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) grow_bits();
        uint32_t k = bitbuf >> (32 - n);
        bitbuf <<= n;
        bitcnt -= n;
        return (int)k;
    }
    int decode(const Huff &h) {
        if (bitcnt < 16) grow_bits();
        uint32_t top = bitbuf >> 16;
        int k;
        for (k = 1; k <= 16; k++)
            if ((int)top < h.maxcode[k]) break;
        if (k == 17) throw std::runtime_error("JPEG: bad Huffman code");
        int idx = (int)((bitbuf >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
        if (idx < 0 || idx >= h.count) throw std::runtime_error("JPEG: bad Huffman code");
        bitbuf <<= k;
        bitcnt -= k;
        return h.values[idx];
    }
    int extend_receive(int n) { // JPEG F.2.2.1
        if (n == 0) return 0;
        int v = get_bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

    // stb_image's stbi__idct_block
    static void idct(uint8_t *out, int stride, const short d[64]) {
#define F2F(x) ((int)(((x) * 4096 + 0.5)))
#define FSH(x) ((x) * 4096)
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                              \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                \
    p2 = s2; p3 = s6;                                                                       \
    p1 = (p2 + p3) * F2F(0.5411961f);                                                       \
    t2 = p1 + p3 * F2F(-1.847759065f);                                                      \
    t3 = p1 + p2 * F2F(0.765366865f);                                                       \
    p2 = s0; p3 = s4;                                                                       \
    t0 = FSH(p2 + p3);                                                                      \
    t1 = FSH(p2 - p3);                                                                      \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                 \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                     \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                 \
    p5 = (p3 + p4) * F2F(1.175875602f);                                                     \
    t0 = t0 * F2F(0.298631336f);                                                            \
    t1 = t1 * F2F(2.053119869f);                                                            \
    t2 = t2 * F2F(3.072711026f);                                                            \
    t3 = t3 * F2F(1.501321110f);                                                            \
    p1 = p5 + p1 * F2F(-0.899976223f);                                                      \
    p2 = p5 + p2 * F2F(-2.562915447f);                                                      \
    p3 = p3 * F2F(-1.961570560f);                                                           \
    p4 = p4 * F2F(-0.390180644f);                                                           \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        int val[64], *v = val;
        const short *dd = d;
        for (int i = 0; i < 8; i++, dd++, v++) {
            if (dd[8] == 0 && dd[16] == 0 && dd[24] == 0 && dd[32] == 0 && dd[40] == 0 && dd[48] == 0 && dd[56] == 0) {
                int dcterm = dd[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                IDCT_1D(dd[0], dd[8], dd[16], dd[24], dd[32], dd[40], dd[48], dd[56])
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
                v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
                v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        v = val;
        uint8_t *o = out;
        for (int i = 0; i < 8; i++, v += 8, o += stride) {
            IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
            x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
#undef IDCT_1D
#undef F2F
#undef FSH
    }

    void decode_block(short data[64], Component &c) {
        static const uint8_t dezigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
        memset(data, 0, 64 * sizeof(short));
        int t = decode(dc[c.hd]);
        if (t > 15) throw std::runtime_error("JPEG: bad DC size");
        int diff = t ? extend_receive(t) : 0;
        c.dc_pred += diff;
        data[0] = (short)(c.dc_pred * dequant[c.tq][0]);
        int k = 1;
        do {
            int rs = decode(ac[c.ha]);
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break; // end of block
                k += 16;
            } else {
                k += r;
                int zig = dezigzag[k++];
                data[zig] = (short)(extend_receive(s) * dequant[c.tq][zig]);
            }
        } while (k < 64);
    }

    void reset_entropy() {
        bitbuf = 0; bitcnt = 0; hit_marker = false; marker = 0;
        for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0;
    }

    void read_tables_until_sos() {
        for (;;) {
            int m = get8();
            while (m != 0xff) { if (p >= end) throw std::runtime_error("JPEG: no start of scan"); m = get8(); }
            while (m == 0xff) m = get8();
            if (m == 0xda) return;
            int len = get16() - 2;
            if (len < 0 || p + len > end) throw std::runtime_error("JPEG: truncated segment");
            const uint8_t *seg_end = p + len;
            if (m == 0xdb) { // DQT
                while (p < seg_end) {
                    int q = get8(), prec = q >> 4, t = q & 15;
                    if (t > 3) throw std::runtime_error("JPEG: bad quantisation table id");
                    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
                    for (int i = 0; i < 64; i++) dequant[t][zz[i]] = (uint16_t)(prec ? get16() : get8());
                }
            } else if (m == 0xc4) { // DHT
                while (p < seg_end) {
                    int q = get8(), tc = q >> 4, th = q & 15;
                    if (tc > 1 || th > 3) throw std::runtime_error("JPEG: bad Huffman table id");
                    int counts[16], n = 0;
                    for (int i = 0; i < 16; i++) { counts[i] = get8(); n += counts[i]; }
                    if (n > 256) throw std::runtime_error("JPEG: bad Huffman table");
                    uint8_t vals[256];
                    for (int i = 0; i < n; i++) vals[i] = (uint8_t)get8();
                    (tc == 0 ? dc[th] : ac[th]).build(counts, vals);
                }
            } else if (m == 0xc0 || m == 0xc1) { // SOF0 / SOF1
                int prec = get8();
                if (prec != 8) throw std::runtime_error("JPEG: only 8-bit precision is supported");
                height = get16(); width = get16();
                ncomp = get8();
                if (width <= 0 || height <= 0) throw std::runtime_error("JPEG: empty image");
                if (ncomp != 1 && ncomp != 3) throw std::runtime_error("JPEG: only grayscale and YCbCr images are supported");
                for (int i = 0; i < ncomp; i++) {
                    comp[i].id = get8();
                    int q = get8();
                    comp[i].h = q >> 4; comp[i].v = q & 15;
                    comp[i].tq = get8();
                    if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw std::runtime_error("JPEG: bad component");
                }
            } else if (m == 0xc2) throw std::runtime_error("JPEG: progressive files are not supported");
            else if (m == 0xdd) restart_interval = get16();
            else if ((m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) throw std::runtime_error("JPEG: unsupported coding process");
            p = seg_end; // APPn, COM and anything else: skipped
        }
    }

    void decode_image(int &w, int &h, std::vector<uint8_t> &rgb) {
        if (get8() != 0xff || get8() != 0xd8) throw std::runtime_error("JPEG: missing SOI");
        read_tables_until_sos();
        if (ncomp == 0) throw std::runtime_error("JPEG: scan before frame header");
        // scan header
        get16();
        int ns = get8();
        if (ns != ncomp) throw std::runtime_error("JPEG: multi-scan (non-interleaved) files are not supported");
        for (int i = 0; i < ns; i++) {
            int id = get8(), q = get8(), which = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == id) which = k;
            if (which != i) throw std::runtime_error("JPEG: unexpected component order in scan");
            comp[i].hd = q >> 4; comp[i].ha = q & 15;
            if (comp[i].hd > 3 || comp[i].ha > 3 || !dc[comp[i].hd].present || !ac[comp[i].ha].present) throw std::runtime_error("JPEG: missing Huffman table");
        }
        get8(); get8(); get8(); // Ss, Se, Ah/Al
        hmax = vmax = 1;
        for (int i = 0; i < ncomp; i++) { if (comp[i].h > hmax) hmax = comp[i].h; if (comp[i].v > vmax) vmax = comp[i].v; }
        for (int i = 0; i < ncomp; i++) if (hmax % comp[i].h || vmax % comp[i].v) throw std::runtime_error("JPEG: fractional sampling ratios are not supported");
        int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
        int mcux = (width + mcu_w - 1) / mcu_w, mcuy = (height + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; i++) {
            comp[i].w2 = mcux * comp[i].h * 8; comp[i].h2 = mcuy * comp[i].v * 8;
            comp[i].data.assign((size_t)comp[i].w2 * comp[i].h2, 0);
        }
        reset_entropy();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        short block[64];
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                for (int c = 0; c < ncomp; c++)
                    for (int by = 0; by < comp[c].v; by++)
                        for (int bx = 0; bx < comp[c].h; bx++) {
                            decode_block(block, comp[c]);
                            int x2 = (mx * comp[c].h + bx) * 8, y2 = (my * comp[c].v + by) * 8;
                            idct(comp[c].data.data() + (size_t)comp[c].w2 * y2 + x2, comp[c].w2, block);
                        }
                if (--todo <= 0) {
                    if (bitcnt < 24) grow_bits();
                    if (!(marker >= 0xd0 && marker <= 0xd7)) goto done; // no restart marker: end of data
                    reset_entropy();
                    todo = restart_interval;
                }
            }
    done:
        // upsample + colour conversion, row by row, as stb_image's load_jpeg_image does
        w = width; h = height;
        rgb.assign((size_t)width * height * 3, 0);
        struct Resample { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
        Resample rs[3];
        for (int k = 0; k < ncomp; k++) {
            rs[k].hs = hmax / comp[k].h; rs[k].vs = vmax / comp[k].v;
            if (!((rs[k].hs == 1 || rs[k].hs == 2) && (rs[k].vs == 1 || rs[k].vs == 2))) throw std::runtime_error("JPEG: only 1x and 2x chroma subsampling is supported");
            rs[k].ystep = rs[k].vs >> 1;
            rs[k].w_lores = (width + rs[k].hs - 1) / rs[k].hs;
            rs[k].ypos = 0;
            rs[k].line0 = rs[k].line1 = comp[k].data.data();
            rs[k].buf.assign((size_t)width + 3, 0);
        }
        for (int j = 0; j < height; j++) {
            const uint8_t *rows[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < ncomp; k++) {
                Resample &r = rs[k];
                bool y_bot = r.ystep >= (r.vs >> 1);
                const uint8_t *in_near = y_bot ? r.line1 : r.line0, *in_far = y_bot ? r.line0 : r.line1;
                uint8_t *out = r.buf.data();
                int wl = r.w_lores;
                if (r.hs == 1 && r.vs == 1) rows[k] = in_near;
                else if (r.hs == 1 && r.vs == 2) { for (int i = 0; i < wl; i++) out[i] = (uint8_t)((3 * in_near[i] + in_far[i] + 2) >> 2); rows[k] = out; }
                else if (r.hs == 2 && r.vs == 1) {
                    const uint8_t *in = in_near;
                    if (wl == 1) out[0] = out[1] = in[0];
                    else {
                        out[0] = in[0];
                        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                        int i;
                        for (i = 1; i < wl - 1; i++) { int n = 3 * in[i] + 2; out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
                        out[i * 2] = (uint8_t)((in[wl - 2] * 3 + in[wl - 1] + 2) >> 2);
                        out[i * 2 + 1] = in[wl - 1];
                    }
                    rows[k] = out;
                } else {
                    if (wl == 1) out[0] = out[1] = (uint8_t)((3 * in_near[0] + in_far[0] + 2) >> 2);
                    else {
                        int t1 = 3 * in_near[0] + in_far[0], t0;
                        out[0] = (uint8_t)((t1 + 2) >> 2);
                        for (int i = 1; i < wl; i++) {
                            t0 = t1;
                            t1 = 3 * in_near[i] + in_far[i];
                            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                        }
                        out[wl * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
                    }
                    rows[k] = out;
                }
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].h2 && r.ypos < (height + r.vs - 1) / r.vs) r.line1 += comp[k].w2;
                }
            }
            uint8_t *o = rgb.data() + (size_t)j * width * 3;
            if (ncomp == 1) for (int i = 0; i < width; i++) { o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = rows[0][i]; }
            else
                for (int i = 0; i < width; i++) { // stbi__YCbCr_to_RGB_row
#define FLOAT2FIXED(x) (((int)((x) * 4096.0f + 0.5f)) << 8)
                    int y_fixed = (rows[0][i] << 20) + (1 << 19);
                    int cr = rows[2][i] - 128, cb = rows[1][i] - 128;
                    int r = y_fixed + cr * FLOAT2FIXED(1.40200f);
                    int g = y_fixed + (cr * -FLOAT2FIXED(0.71414f)) + ((cb * -FLOAT2FIXED(0.34414f)) & 0xffff0000);
                    int b = y_fixed + cb * FLOAT2FIXED(1.77200f);
#undef FLOAT2FIXED
                    r >>= 20; g >>= 20; b >>= 20;
                    o[3 * i] = clamp8(r); o[3 * i + 1] = clamp8(g); o[3 * i + 2] = clamp8(b);
                }
        }
    }
};

} // namespace

void decode_jpeg(const std::vector<uint8_t> &file, int &width, int &height, std::vector<uint8_t> &rgb) {
    Decoder d(file.data(), file.data() + file.size());
    d.decode_image(width, height, rgb);
}

} // namespace rtamd
