// rtx_image.cpp — texture files -> the float3 texel arrays the render path samples (CPU, load time).
//
//   rtxh_texture_load   Texture::load                                   Texture.cpp:30-129
//                       (colour_unpack :13-20, Math::gamma_to_linear Math.h:67-77, box-filter mips :76-117)
//   rtxh_image_load     what Texture::load gets back from its image library: stbi_load(path, &w, &h, &n, STBI_rgb_alpha)
//                       (Texture.cpp:40; stb_image v2.19 is vendored by the reference under include/stb_image).
//
// The decoder is written from the PNG (RFC 2083 / zlib RFC 1950-1951) and Truevision TGA format definitions, with the
// choices stb_image makes where the formats leave room, because those choices reach the texels:
//   * PNG 16-bit samples are reduced to their high byte; 1/2/4-bit grey is scaled by 255/85/17; palettes are expanded;
//     gAMA / sRGB / iCCP chunks are ignored; CRCs and the Adler-32 are not verified;
//   * TGA 15/16-bit pixels expand as (c * 255) / 31; bit 5 of the descriptor selects top-down rows, bit 4 (right-to-left)
//     is ignored; colour-mapped, grey and RLE variants are handled; out-of-range palette indices read entry 0;
//   * the file type is probed PNG first and TGA last; JPEG/BMP/GIF/PSD/PIC/PNM/HDR (which stb_image would try in
//     between) are reported as unsupported — the reference's scenes reference only .png and .tga textures.
// tests/test_image.py compares decoded bytes with those of the reference's own stb_image (oracle/_ref/stbdump) on the
// reference's data files and on synthetic files covering every PNG colour type / bit depth / interlacing and TGA type.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/rtx_host.h"

namespace {

typedef std::vector<uint8_t> Bytes;

// ---------------------------------------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
    const uint8_t * p; size_t n, pos = 0; uint32_t acc = 0; int bits = 0; bool bad = false;
    BitReader(const uint8_t * d, size_t len) : p(d), n(len) {}
    uint32_t get(int k) {
        while (bits < k) { uint32_t b = 0; if (pos < n) b = p[pos++]; else bad = true; acc |= b << bits; bits += 8; }
        uint32_t v = acc & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u));
        acc >>= k; bits -= k;
        return v;
    }
    void align() { acc = 0; bits = 0; }
};

struct Huffman {
    uint16_t count[16]; uint16_t symbol[288];
    bool build(const uint8_t * lengths, int n) {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader & br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.get(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
            if (br.bad) return -1;
        }
        return -1;
    }
};

bool inflate_zlib(const uint8_t * data, size_t len, Bytes & out, bool has_header, size_t max_out) {
    // max_out: the caller knows how many bytes it can use; decoding stops there (a few IDAT bytes must not inflate to gigabytes)
    static const uint16_t LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    static const uint8_t CL_ORDER[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    BitReader br(data, len);
    if (has_header) {
        uint32_t cmf = br.get(8), flg = br.get(8);
        if (br.bad || (cmf * 256 + flg) % 31 != 0 || (flg & 32) || (cmf & 15) != 8) return false;
    }
    for (;;) {
        uint32_t final = br.get(1), type = br.get(2);
        if (br.bad) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > len) return false;
            uint32_t n = data[br.pos] | (data[br.pos + 1] << 8), nn = data[br.pos + 2] | (data[br.pos + 3] << 8);
            br.pos += 4;
            if ((n ^ 0xffffu) != nn || br.pos + n > len) return false;
            out.insert(out.end(), data + br.pos, data + br.pos + n);
            br.pos += n;
            if (out.size() >= max_out) return true;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lengths[i] = 8;
                for (int i = 144; i < 256; i++) lengths[i] = 9;
                for (int i = 256; i < 280; i++) lengths[i] = 7;
                for (int i = 280; i < 288; i++) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 32; i++) lengths[i] = 5;
                dist.build(lengths, 32);
            } else {
                int hlit = (int)br.get(5) + 257, hdist = (int)br.get(5) + 1, hclen = (int)br.get(4) + 4;
                if (hlit > 286 || hdist > 32) return false;      // stb_image accepts up to 32 distance codes
                uint8_t cl[19] = { 0 };
                for (int i = 0; i < hclen; i++) cl[CL_ORDER[i]] = (uint8_t)br.get(3);
                Huffman clh;
                if (!clh.build(cl, 19)) return false;
                int i = 0;
                while (i < hlit + hdist) {
                    int s = clh.decode(br);
                    if (s < 0) return false;
                    if (s < 16) lengths[i++] = (uint8_t)s;
                    else {
                        uint8_t fill = 0; int rep;
                        if (s == 16) { if (i == 0) return false; fill = lengths[i - 1]; rep = 3 + (int)br.get(2); }
                        else if (s == 17) rep = 3 + (int)br.get(3);
                        else rep = 11 + (int)br.get(7);
                        if (i + rep > hlit + hdist) return false;
                        while (rep--) lengths[i++] = fill;
                    }
                }
                if (!lit.build(lengths, hlit)) return false;
                if (!dist.build(lengths + hlit, hdist)) return false;
            }
            for (;;) {
                int s = lit.decode(br);
                if (s < 0 || br.bad) return false;
                if (s < 256) { out.push_back((uint8_t)s); if (out.size() >= max_out) return true; }
                else if (s == 256) break;
                else {
                    s -= 257;
                    if (s >= 29) return false;
                    int length = LEN_BASE[s] + (int)br.get(LEN_EXTRA[s]);
                    int d = dist.decode(br);
                    if (d < 0 || d >= 30) return false;
                    size_t distance = DIST_BASE[d] + br.get(DIST_EXTRA[d]);
                    if (distance > out.size()) return false;
                    size_t from = out.size() - distance;
                    for (int k = 0; k < length; k++) out.push_back(out[from + k]);
                    if (out.size() >= max_out) return true;
                }
            }
        } else return false;
        if (final) break;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- PNG
inline uint32_t be32(const uint8_t * p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

// Undo the row filters of one (sub)image of w x h pixels; `raw` holds h rows of 1 + stride bytes.  Returns samples
// widened to one uint16 per sample (value as stored: depth bits), channels interleaved.
bool png_unfilter(const uint8_t * raw, size_t raw_len, int w, int h, int channels, int depth, std::vector<uint16_t> & samples) {
    const size_t stride = ((size_t)w * channels * depth + 7) / 8;
    const int bpp = (channels * depth + 7) / 8;                       // filter unit, at least one byte
    if (raw_len < (stride + 1) * (size_t)h) return false;
    Bytes prev(stride, 0), cur(stride);
    samples.resize((size_t)w * h * channels);
    for (int y = 0; y < h; y++) {
        const uint8_t * row = raw + (stride + 1) * (size_t)y;
        const int filter = row[0];
        if (filter > 4) return false;
        for (size_t i = 0; i < stride; i++) {
            const int x = row[1 + i];
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v;
            switch (filter) {
                case 0: v = x; break;
                case 1: v = x + a; break;
                case 2: v = x + b; break;
                case 3: v = x + ((a + b) >> 1); break;
                default: v = x + paeth(a, b, c); break;
            }
            cur[i] = (uint8_t)v;
        }
        uint16_t * o = samples.data() + (size_t)y * w * channels;
        const int n = w * channels;
        if (depth == 8) for (int i = 0; i < n; i++) o[i] = cur[i];
        else if (depth == 16) for (int i = 0; i < n; i++) o[i] = (uint16_t)((cur[2 * i] << 8) | cur[2 * i + 1]);
        else for (int i = 0; i < n; i++) { const int bit = i * depth; o[i] = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1); }
        prev.swap(cur);
    }
    return true;
}

int png_decode(const Bytes & f, int32_t * w_out, int32_t * h_out, Bytes & rgba) {
    size_t pos = 8;
    uint32_t w = 0, h = 0; int depth = 0, color = 0, interlace = 0;
    bool have_ihdr = false, is_iphone = false, has_key = false;
    uint8_t palette[256][4]; int pal_len = 0;
    uint16_t key[3] = { 0, 0, 0 };
    Bytes idat;
    for (;;) {
        if (pos + 8 > f.size()) return RTX_ERR_INVALID_ARG;
        const uint32_t len = be32(&f[pos]); const uint32_t type = be32(&f[pos + 4]);
        const uint8_t * d = f.data() + pos + 8;
        if (pos + 8 + (size_t)len > f.size()) return RTX_ERR_INVALID_ARG;                        // truncated file
        if (type == 0x43674249u) is_iphone = true;                                               // CgBI
        else if (type == 0x49484452u) {                                                          // IHDR
            if (have_ihdr || len != 13) return RTX_ERR_INVALID_ARG;
            have_ihdr = true;
            w = be32(d); h = be32(d + 4); depth = d[8]; color = d[9]; interlace = d[12];
            if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) return RTX_ERR_LIMIT;
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return RTX_ERR_INVALID_ARG;
            if (color > 6 || color == 1 || color == 5 || (color == 3 && depth == 16)) return RTX_ERR_INVALID_ARG;
            if (d[10] || d[11] || interlace > 1) return RTX_ERR_INVALID_ARG;
            if ((1u << 30) / w / 4 < h) return RTX_ERR_LIMIT;
        } else if (!have_ihdr) return RTX_ERR_INVALID_ARG;
        else if (type == 0x504c5445u) {                                                          // PLTE
            if (len > 768 || len % 3) return RTX_ERR_INVALID_ARG;
            pal_len = (int)(len / 3);
            for (int i = 0; i < pal_len; i++) { palette[i][0] = d[3 * i]; palette[i][1] = d[3 * i + 1]; palette[i][2] = d[3 * i + 2]; palette[i][3] = 255; }
        } else if (type == 0x74524e53u) {                                                        // tRNS
            if (!idat.empty()) return RTX_ERR_INVALID_ARG;
            if (color == 3) {
                if (pal_len == 0 || (int)len > pal_len) return RTX_ERR_INVALID_ARG;
                for (uint32_t i = 0; i < len; i++) palette[i][3] = d[i];
            } else {
                const int n = (color & 2) ? 3 : 1;
                if ((color & 4) || len != (uint32_t)n * 2) return RTX_ERR_INVALID_ARG;
                has_key = true;
                for (int k = 0; k < n; k++) key[k] = (uint16_t)((d[2 * k] << 8) | d[2 * k + 1]);
            }
        } else if (type == 0x49444154u) {                                                        // IDAT
            if (color == 3 && pal_len == 0) return RTX_ERR_INVALID_ARG;
            idat.insert(idat.end(), d, d + len);
        } else if (type == 0x49454e44u) break;                                                   // IEND
        else if ((type & (1u << 29)) == 0) return RTX_ERR_INVALID_ARG;                           // unknown critical chunk
        pos += 12 + (size_t)len;
    }
    if (idat.empty()) return RTX_ERR_INVALID_ARG;
    const int channels = color == 3 ? 1 : ((color & 2) ? 3 : 1) + ((color & 4) ? 1 : 0);
    static const int X0[7] = { 0, 4, 0, 2, 0, 1, 0 }, Y0[7] = { 0, 0, 4, 0, 2, 0, 1 }, DX[7] = { 8, 8, 4, 4, 2, 2, 1 }, DY[7] = { 8, 8, 8, 4, 4, 2, 2 };
    size_t expected = 0;                                                                          // filtered bytes the image needs
    if (!interlace) expected = ((((size_t)w * channels * depth + 7) / 8) + 1) * h;
    else for (int p = 0; p < 7; p++) {
        const long pw = ((long)w - X0[p] + DX[p] - 1) / DX[p], ph = ((long)h - Y0[p] + DY[p] - 1) / DY[p];
        if (pw > 0 && ph > 0) expected += ((((size_t)pw * channels * depth + 7) / 8) + 1) * (size_t)ph;
    }
    Bytes raw;
    if (!inflate_zlib(idat.data(), idat.size(), raw, !is_iphone, expected)) return RTX_ERR_INVALID_ARG;
    if (raw.size() < expected) return RTX_ERR_INVALID_ARG;                                        // checked before anything image-sized is allocated

    std::vector<uint16_t> img((size_t)w * h * channels);
    if (!interlace) {
        if (!png_unfilter(raw.data(), raw.size(), (int)w, (int)h, channels, depth, img)) return RTX_ERR_INVALID_ARG;
    } else {                                                                                     // Adam7
        size_t off = 0;
        std::vector<uint16_t> pass;
        for (int p = 0; p < 7; p++) {
            const int pw = ((int)w - X0[p] + DX[p] - 1) / DX[p], ph = ((int)h - Y0[p] + DY[p] - 1) / DY[p];
            if (pw <= 0 || ph <= 0) continue;
            const size_t bytes = ((((size_t)pw * channels * depth + 7) / 8) + 1) * ph;
            if (off + bytes > raw.size()) return RTX_ERR_INVALID_ARG;
            if (!png_unfilter(raw.data() + off, bytes, pw, ph, channels, depth, pass)) return RTX_ERR_INVALID_ARG;
            off += bytes;
            for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++)
                memcpy(&img[(((size_t)(Y0[p] + y * DY[p])) * w + X0[p] + x * DX[p]) * channels], &pass[((size_t)y * pw + x) * channels], sizeof(uint16_t) * channels);
        }
    }

    rgba.resize((size_t)w * h * 4);
    static const int SCALE[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    const size_t px = (size_t)w * h;
    for (size_t i = 0; i < px; i++) {
        const uint16_t * s = &img[i * channels];
        uint8_t * o = &rgba[4 * i];
        if (color == 3) {
            const int idx = s[0];
            if (idx < pal_len) memcpy(o, palette[idx], 4);
            else { o[0] = o[1] = o[2] = 0; o[3] = 255; }             // past the PLTE chunk stb_image reads uninitialised stack; black here
            continue;
        }
        uint8_t v[4]; bool transparent = false;
        if (depth == 16) {
            if (has_key) { transparent = true; for (int k = 0; k < ((color & 2) ? 3 : 1); k++) transparent = transparent && s[k] == key[k]; }
            for (int k = 0; k < channels; k++) v[k] = (uint8_t)(s[k] >> 8);
        } else {
            for (int k = 0; k < channels; k++) v[k] = (uint8_t)(s[k] * SCALE[depth]);
            if (has_key) { transparent = true; for (int k = 0; k < ((color & 2) ? 3 : 1); k++) transparent = transparent && v[k] == (uint8_t)((uint8_t)(key[k] & 255) * SCALE[depth]); }
        }
        if (color & 2) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = (color & 4) ? v[3] : 255; }
        else           { o[0] = o[1] = o[2] = v[0];             o[3] = (color & 4) ? v[1] : 255; }
        if (has_key) o[3] = transparent ? 0 : 255;
    }
    *w_out = (int32_t)w; *h_out = (int32_t)h;
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------- TGA
struct TgaHeader { int id_len, indexed, type, pal_start, pal_len, pal_bits, w, h, bpp, descriptor; bool rle; };

int tga_components(int bits, bool grey, bool * rgb16) {
    *rgb16 = false;
    switch (bits) {
        case 8: return 1;
        case 16: if (grey) return 2;   /* fall through */
        case 15: *rgb16 = true; return 3;
        case 24: case 32: return bits / 8;
        default: return 0;
    }
}

bool tga_probe(const Bytes & f, TgaHeader & t) {
    if (f.size() < 18) return false;
    const uint8_t * d = f.data();
    t.id_len = d[0]; t.indexed = d[1]; t.type = d[2];
    t.pal_start = d[3] | (d[4] << 8); t.pal_len = d[5] | (d[6] << 8); t.pal_bits = d[7];
    t.w = d[12] | (d[13] << 8); t.h = d[14] | (d[15] << 8); t.bpp = d[16]; t.descriptor = d[17];
    if (t.indexed > 1) return false;
    bool rgb16;
    if (t.indexed == 1) {
        if (t.type != 1 && t.type != 9) return false;
        if (t.pal_bits != 8 && t.pal_bits != 15 && t.pal_bits != 16 && t.pal_bits != 24 && t.pal_bits != 32) return false;
        if (t.bpp != 8 && t.bpp != 16) return false;
        if (!tga_components(t.pal_bits, false, &rgb16)) return false;
    } else {
        if (t.type != 2 && t.type != 3 && t.type != 10 && t.type != 11) return false;
        if (!tga_components(t.bpp, t.type == 3 || t.type == 11, &rgb16)) return false;
    }
    if (t.w < 1 || t.h < 1) return false;
    t.rle = t.type >= 8;
    if (t.rle) t.type -= 8;
    return true;
}

int tga_decode(const Bytes & f, const TgaHeader & t, int32_t * w_out, int32_t * h_out, Bytes & rgba) {
    bool rgb16;
    const int comp = t.indexed ? tga_components(t.pal_bits, false, &rgb16) : tga_components(t.bpp, t.type == 3, &rgb16);
    if (!comp) return RTX_ERR_INVALID_ARG;
    size_t pos = 18 + (size_t)t.id_len;
    auto get8 = [&]() -> int { return pos < f.size() ? f[pos++] : (pos++, 0); };                 // reads past the end give zeros, as stb's do
    auto get16 = [&]() -> int { int a = get8(); return a | (get8() << 8); };
    auto rgb16_px = [&](uint8_t * o) { int px = get16(); o[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31); o[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31); o[2] = (uint8_t)(((px & 31) * 255) / 31); };

    const size_t px_count = (size_t)t.w * t.h;
    if (px_count * comp > 0x7fffffffull) return RTX_ERR_LIMIT;                                   // stb_image's own limit (sizes must fit an int)
    {   // a header can claim 65535 x 65535 pixels on a 20-byte file; stb_image would then read zeros past the end for minutes.  Large
        // images must at least be long enough to hold their smallest possible encoding (RLE: one packet per 128 pixels).
        const size_t remaining = f.size() > pos ? f.size() - pos : 0, unit = (size_t)(t.bpp + 7) / 8;
        const size_t smallest = t.rle ? ((px_count + 127) / 128) * (1 + unit) : px_count * unit;
        if (px_count > (1u << 22) && remaining < smallest) return RTX_ERR_INVALID_ARG;
    }
    Bytes data(px_count * comp);
    Bytes pal;
    if (t.indexed) {
        pos += (size_t)t.pal_start;
        pal.resize((size_t)t.pal_len * comp + 4);
        for (int i = 0; i < t.pal_len; i++) {
            if (rgb16) rgb16_px(&pal[(size_t)i * comp]);
            else for (int k = 0; k < comp; k++) pal[(size_t)i * comp + k] = (uint8_t)get8();
        }
    }
    uint8_t cur[4] = { 0, 0, 0, 0 };
    int run = 0; bool repeating = false, read_next = true;
    for (size_t i = 0; i < px_count; i++) {
        if (t.rle) {
            if (run == 0) { int cmd = get8(); run = 1 + (cmd & 127); repeating = (cmd >> 7) != 0; read_next = true; }
            else if (!repeating) read_next = true;
        } else read_next = true;
        if (read_next) {
            if (t.indexed) {
                int idx = t.bpp == 8 ? get8() : get16();
                if (idx >= t.pal_len) idx = 0;
                for (int k = 0; k < comp; k++) cur[k] = pal[(size_t)idx * comp + k];
            } else if (rgb16) rgb16_px(cur);
            else for (int k = 0; k < comp; k++) cur[k] = (uint8_t)get8();
            read_next = false;
        }
        memcpy(&data[i * comp], cur, comp);
        run--;
    }
    const bool bottom_up = ((t.descriptor >> 5) & 1) == 0;
    rgba.resize(px_count * 4);
    for (int y = 0; y < t.h; y++) {
        const int sy = bottom_up ? t.h - 1 - y : y;
        for (int x = 0; x < t.w; x++) {
            const uint8_t * s = &data[((size_t)sy * t.w + x) * comp];
            uint8_t * o = &rgba[((size_t)y * t.w + x) * 4];
            if (comp == 1)      { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
            else if (comp == 2) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
            else if (rgb16)     { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; }
            else                { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = comp == 4 ? s[3] : 255; }   // stored B, G, R(, A)
        }
    }
    *w_out = t.w; *h_out = t.h;
    return RTX_OK;
}

bool read_file(const char * path, Bytes & out) {
    FILE * f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int decode_any(const Bytes & f, int32_t * w, int32_t * h, Bytes & rgba) {
    static const uint8_t PNG_SIG[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (f.size() >= 8 && memcmp(f.data(), PNG_SIG, 8) == 0) return png_decode(f, w, h, rgba);
    // the loaders stb_image tries between PNG and TGA: recognise, refuse
    if (f.size() >= 2 && f[0] == 0xff && f[1] == 0xd8) return RTX_ERR_LIMIT;                                     // JPEG
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return RTX_ERR_LIMIT;                                       // BMP
    if (f.size() >= 6 && (memcmp(f.data(), "GIF87a", 6) == 0 || memcmp(f.data(), "GIF89a", 6) == 0)) return RTX_ERR_LIMIT;
    if (f.size() >= 4 && memcmp(f.data(), "8BPS", 4) == 0) return RTX_ERR_LIMIT;                                 // PSD
    if (f.size() >= 4 && f[0] == 0x53 && f[1] == 0x80 && f[2] == 0xf6 && f[3] == 0x34) return RTX_ERR_LIMIT;     // PIC
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return RTX_ERR_LIMIT;                      // PNM
    if (f.size() >= 10 && (memcmp(f.data(), "#?RADIANCE", 10) == 0 || memcmp(f.data(), "#?RGBE", 6) == 0)) return RTX_ERR_LIMIT;
    TgaHeader t;
    if (tga_probe(f, t)) return tga_decode(f, t, w, h, rgba);
    return RTX_ERR_INVALID_ARG;
}

}  // namespace

extern "C" int rtxh_image_load(const char * path, int32_t * width, int32_t * height, uint8_t ** rgba_out) {
    if (!path || !width || !height || !rgba_out) return RTX_ERR_INVALID_ARG;
    Bytes file, rgba;
    if (!read_file(path, file)) return RTX_ERR_STATE;
    int rc = decode_any(file, width, height, rgba);
    if (rc) return rc;
    uint8_t * out = (uint8_t *)malloc(rgba.size());
    if (!out) return RTX_ERR_OOM;
    memcpy(out, rgba.data(), rgba.size());
    *rgba_out = out;
    return RTX_OK;
}

extern "C" int rtxh_image_free(uint8_t * rgba) { free(rgba); return RTX_OK; }

// A frame (Window::frame_buffer layout, 0x00RRGGBB) as an 8-bit RGB PNG: filter 0 rows in stored (uncompressed) deflate blocks.
extern "C" int rtxh_image_save_png(const char * path, const uint32_t * packed, int32_t width, int32_t height) {
    if (!path || !packed || width <= 0 || height <= 0) return RTX_ERR_INVALID_ARG;
    static uint32_t crc_table[256]; static bool crc_ready = false;
    if (!crc_ready) { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; } crc_ready = true; }
    auto crc = [&](const uint8_t * d, size_t n, uint32_t c) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ d[i]) & 255] ^ (c >> 8); return c; };
    Bytes raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        for (int x = 0; x < width; x++) { const uint32_t p = packed[(size_t)y * width + x]; raw.push_back((uint8_t)(p >> 16)); raw.push_back((uint8_t)(p >> 8)); raw.push_back((uint8_t)p); }
    }
    Bytes z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        if (off + n >= raw.size()) break;
    }
    const uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; k--) z.push_back((uint8_t)(adler >> (8 * k)));
    FILE * f = fopen(path, "wb");
    if (!f) return RTX_ERR_STATE;
    auto chunk = [&](const char * tag, const uint8_t * d, size_t n) {
        uint8_t len[4] = { (uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n };
        fwrite(len, 1, 4, f); fwrite(tag, 1, 4, f); if (n) fwrite(d, 1, n, f);
        uint32_t c = crc((const uint8_t *)tag, 4, 0xffffffffu); c = crc(d, n, c) ^ 0xffffffffu;
        uint8_t cb[4] = { (uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c };
        fwrite(cb, 1, 4, f);
    };
    static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = { (uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                         (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height, 8, 2, 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? RTX_OK : RTX_ERR_STATE;
}

extern "C" int rtxh_texture_load(const char * path, int32_t mipmap_mode, float ** texels_out, int64_t * texel_count_out, rtx_texture_desc * desc) {
    if (!path || !texels_out || !texel_count_out || !desc) return RTX_ERR_INVALID_ARG;
    int32_t w = 0, h = 0;
    Bytes file, rgba;
    if (!read_file(path, file)) return RTX_ERR_STATE;
    int rc = decode_any(file, &w, &h, rgba);
    if (rc) return rc;
    float lut[256];                                                     // colour_unpack + Math::gamma_to_linear, one evaluation per byte value
    for (int b = 0; b < 256; b++) {
        const float x = float(b) * 0.00392156862f;
        lut[b] = x <= 0.0f ? 0.0f : x >= 1.0f ? 1.0f : x < 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f);
    }
    const size_t px = (size_t)w * h;
    float * tx = (float *)malloc(sizeof(float) * 3 * (px + px / 3 + 2));
    if (!tx) return RTX_ERR_OOM;
    for (size_t i = 0; i < px; i++) for (int k = 0; k < 3; k++) tx[3 * i + k] = lut[rgba[4 * i + k]];
    const bool pow2 = ((w & (w - 1)) == 0) && ((h & (h - 1)) == 0);
    if (mipmap_mode && pow2) {                                          // Texture.cpp:49-55
        rc = rtxh_texture_mips(tx, w, h, desc, texel_count_out);
        if (rc) { free(tx); return rc; }
    } else {
        memset(desc, 0, sizeof(*desc));
        desc->width = w; desc->height = h; desc->mipmapped = 0; desc->mip_levels = 1;
        *texel_count_out = (int64_t)px;
    }
    *texels_out = tx;
    return RTX_OK;
}

extern "C" int rtxh_texture_free(float * texels) { free(texels); return RTX_OK; }

// Sky::Sky (Sky.cpp:8-26): a headerless file of float3 texels of a square angular-map probe (Debevec `.float` files)
extern "C" int rtxh_sky_load(const char * path, float ** texels_out, int32_t * size_out) {
    if (!path || !texels_out || !size_out) return RTX_ERR_INVALID_ARG;
    Bytes file;
    if (!read_file(path, file)) return RTX_ERR_STATE;                    // the reference aborts
    const size_t texels = file.size() / 12;                              // ftell / sizeof(Vector3)
    const int size = (int)sqrtf((float)(int)texels);                     // int(sqrtf(size_squared)), :17
    if (texels == 0 || texels > 0x7fffffffu || (size_t)size * size != texels) return RTX_ERR_INVALID_ARG;   // assert(size * size == size_squared)
    float * out = (float *)malloc(texels * 12);
    if (!out) return RTX_ERR_OOM;
    memcpy(out, file.data(), texels * 12);
    *texels_out = out; *size_out = size;
    return RTX_OK;
}
