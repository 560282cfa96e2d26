// PNG encode (stored deflate blocks) and decode (inflate + unfilter, 8-bit non-interlaced) for the host's output
// image and environment maps.  The reference vendors lodepng for both (Renderer.cpp:104, Scene.hpp:41).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace png_min {

inline uint32_t crc32(const uint8_t *d, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

inline void put32(std::vector<uint8_t> &o, uint32_t v) {
    o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v);
}

inline void chunk(std::vector<uint8_t> &out, const char *tag, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(body.data(), body.size()));
}

// rgba: w*h*4 bytes.  Returns an empty string on success, an error text otherwise.
inline std::string encode_rgba(const std::string &path, const std::vector<uint8_t> &rgba, unsigned w, unsigned h) {
    if (rgba.size() != (size_t)w * h * 4) return "image size mismatch";
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + (size_t)w * 4));
    for (unsigned y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba.begin() + (size_t)y * w * 4, rgba.begin() + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521;
            b = (b + a) % 65521;
        }
        pos += n;
        if (last) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return "failed to open file for writing";
    const size_t wr = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return wr == out.size() ? "" : "short write";
}

// ---- inflate (RFC 1951)
struct BitReader {
    const uint8_t *d;
    size_t n, pos = 0;
    uint32_t bit = 0;
    int bits(int c) {
        int v = 0;
        for (int i = 0; i < c; ++i) {
            if (pos >= n) throw 1;
            v |= ((d[pos] >> bit) & 1) << i;
            if (++bit == 8) { bit = 0; ++pos; }
        }
        return v;
    }
};
struct Huff {
    uint16_t count[16] = {0}, symbol[320] = {0};
    void build(const uint8_t *len, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= br.bits(1);
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw 2;
    }
};
// `limit`: the decoded size the caller expects (the filtered scanlines of the IHDR's image).  The output never grows beyond it: a small
// crafted IDAT cannot expand into gigabytes before the unfilter step rejects it.
inline bool inflate(const std::vector<uint8_t> &z, std::vector<uint8_t> &out, size_t limit) {
    static const uint16_t lbase[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (z.size() < 6) return false;
    BitReader br{z.data() + 2, z.size() - 2};
    try {
        int last;
        do {
            last = br.bits(1);
            const int type = br.bits(2);
            if (type == 0) {
                if (br.bit) { br.bit = 0; ++br.pos; }
                if (br.pos + 4 > br.n) return false;
                const unsigned len = br.d[br.pos] | (br.d[br.pos + 1] << 8);
                br.pos += 4;
                if (br.pos + len > br.n || out.size() + len > limit) return false;
                out.insert(out.end(), br.d + br.pos, br.d + br.pos + len);
                br.pos += len;
            } else if (type == 1 || type == 2) {
                Huff lit, dist;
                uint8_t lens[320];
                if (type == 1) {
                    for (int i = 0; i < 144; ++i) lens[i] = 8;
                    for (int i = 144; i < 256; ++i) lens[i] = 9;
                    for (int i = 256; i < 280; ++i) lens[i] = 7;
                    for (int i = 280; i < 288; ++i) lens[i] = 8;
                    lit.build(lens, 288);
                    for (int i = 0; i < 30; ++i) lens[i] = 5;
                    dist.build(lens, 30);
                } else {
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    const int nl = br.bits(5) + 257, nd = br.bits(5) + 1, nc = br.bits(4) + 4;
                    uint8_t cl[19] = {0};
                    for (int i = 0; i < nc; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                    Huff ch;
                    ch.build(cl, 19);
                    int i = 0;
                    while (i < nl + nd) {
                        const int sym = ch.decode(br);
                        if (sym < 16) lens[i++] = (uint8_t)sym;
                        else {
                            int rep, val = 0;
                            if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + br.bits(2); }
                            else if (sym == 17) rep = 3 + br.bits(3);
                            else rep = 11 + br.bits(7);
                            if (i + rep > nl + nd) return false;
                            while (rep--) lens[i++] = (uint8_t)val;
                        }
                    }
                    lit.build(lens, nl);
                    dist.build(lens + nl, nd);
                }
                while (true) {
                    const int sym = lit.decode(br);
                    if (sym < 256) {
                        if (out.size() >= limit) return false;
                        out.push_back((uint8_t)sym);
                    } else if (sym == 256) break;
                    else {
                        if (sym - 257 >= 29) return false;
                        const int len = lbase[sym - 257] + br.bits(lext[sym - 257]);
                        const int ds = dist.decode(br);
                        if (ds >= 30) return false;
                        const size_t d = dbase[ds] + br.bits(dext[ds]);
                        if (d > out.size() || out.size() + (size_t)len > limit) return false;
                        for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
                    }
                }
            } else {
                return false;
            }
        } while (!last);
    } catch (int) {
        return false;
    }
    return true;
}

// Undoes the scanline filters of one (sub-)image of `h` rows of `stride` bytes (bpp = bytes per complete pixel, at least 1).
inline bool unfilter(const uint8_t *raw, size_t raw_size, unsigned h, size_t stride, int bpp, std::vector<uint8_t> &img) {
    if (raw_size < (size_t)h * (stride + 1)) return false;
    img.assign((size_t)h * stride, 0);
    std::vector<uint8_t> zero(stride, 0);
    for (unsigned y = 0; y < h; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t *line = &raw[y * (stride + 1) + 1];
        uint8_t *cur = &img[y * stride];
        const uint8_t *prev = y ? &img[(y - 1) * stride] : zero.data();
        if (ft > 4) return false;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev[x], cc = x >= (size_t)bpp ? prev[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - cc, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - cc);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : cc);
            }
            cur[x] = (uint8_t)(line[x] + pred);
        }
    }
    return true;
}

// Decodes any PNG into RGBA8, converting as the reference's lodepng::decode(.., LCT_RGBA, 8) does (Scene.hpp:41): grey and palette
// images at 1/2/4/8 bits (grey scaled by 255 / max, palette through PLTE with tRNS alpha), 16-bit samples reduced to their high
// byte, colour keys (tRNS) giving alpha 0, Adam7-interlaced files.  Returns an error text or "".
inline std::string decode_rgba(const std::string &path, std::vector<uint8_t> &rgba, unsigned &w, unsigned &h) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return "failed to open file for reading";
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) return "incorrect PNG signature";
    std::vector<uint8_t> z, plte, trns;
    int depth = 0, ctype = -1, interlace = 0;
    w = h = 0;
    size_t pos = 8;
    while (pos + 12 <= data.size()) {
        const uint32_t len = ((uint32_t)data[pos] << 24) | (data[pos + 1] << 16) | (data[pos + 2] << 8) | data[pos + 3];
        const std::string tag((const char *)&data[pos + 4], 4);
        if (len > data.size() || pos + 12 + len > data.size()) return "truncated chunk";
        const uint8_t *body = &data[pos + 8];
        if (tag == "IHDR" && len >= 13) {
            w = ((uint32_t)body[0] << 24) | (body[1] << 16) | (body[2] << 8) | body[3];
            h = ((uint32_t)body[4] << 24) | (body[5] << 16) | (body[6] << 8) | body[7];
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (tag == "PLTE") {
            plte.assign(body, body + len);
        } else if (tag == "tRNS") {
            trns.assign(body, body + len);
        } else if (tag == "IDAT") {
            z.insert(z.end(), body, body + len);
        } else if (tag == "IEND") {
            break;
        }
        pos += 12 + len;
    }
    const int c = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (c == 0 || !depth_ok || interlace > 1) return "unsupported PNG layout";
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return "image dimensions out of range";  // 268 M pixels: 1 GiB of RGBA
    if (ctype == 3 && plte.size() < 3) return "palette image without PLTE";
    try {
        std::vector<uint8_t> raw;
        const int bits = c * depth;              // bits per pixel
        // what the zlib stream may decode to: one filter byte + the packed row, per row (per Adam7 pass when interlaced)
        size_t expect = 0;
        if (!interlace) {
            expect = (size_t)h * (((size_t)w * bits + 7) / 8 + 1);
        } else {
            static const unsigned ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
            for (int p = 0; p < 7; ++p) {
                const size_t pw = w > ax0[p] ? (w - ax0[p] + adx[p] - 1) / adx[p] : 0, ph = h > ay0[p] ? (h - ay0[p] + ady[p] - 1) / ady[p] : 0;
                if (pw && ph) expect += ph * ((pw * bits + 7) / 8 + 1);
            }
        }
        if (!inflate(z, raw, expect)) return "corrupt zlib stream";
        const int bpp = std::max(1, bits / 8);   // filter distance
        rgba.assign((size_t)w * h * 4, 255);
        // writes pixel (x, y) from sample position `i` of an unfiltered row
        auto put = [&](const uint8_t *row, size_t i, unsigned x, unsigned y) {
            uint8_t *o = &rgba[((size_t)y * w + x) * 4];
            auto sample16 = [&](size_t k) { return (unsigned)(row[2 * k] << 8 | row[2 * k + 1]); };
            if (ctype == 0 || ctype == 3) {
                unsigned v;
                if (depth == 16) v = sample16(i);
                else if (depth == 8) v = row[i];
                else v = (row[(i * depth) >> 3] >> (8 - depth - ((i * depth) & 7))) & ((1u << depth) - 1u);
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 < plte.size()) { o[0] = plte[3 * v]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v + 2]; }
                    else { o[0] = o[1] = o[2] = 0; }
                    o[3] = v < trns.size() ? trns[v] : 255;
                } else {
                    const bool keyed = trns.size() >= 2 && v == (unsigned)(trns[0] << 8 | trns[1]);
                    o[0] = o[1] = o[2] = depth == 16 ? (uint8_t)(v >> 8) : depth == 8 ? (uint8_t)v : (uint8_t)((v * 255u) / ((1u << depth) - 1u));
                    o[3] = keyed ? 0 : 255;
                }
            } else if (depth == 8) {
                const uint8_t *p = &row[i * c];
                if (c == 2) { o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; }
                else {
                    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
                    o[3] = c == 4 ? p[3] : ((trns.size() >= 6 && p[0] == trns[1] && p[1] == trns[3] && p[2] == trns[5] && !trns[0] && !trns[2] && !trns[4]) ? 0 : 255);
                }
            } else {  // 16-bit: the high byte (lodepng's reduction)
                const size_t k = i * c;
                if (c == 2) { o[0] = o[1] = o[2] = row[2 * k]; o[3] = row[2 * k + 2]; }
                else {
                    o[0] = row[2 * k]; o[1] = row[2 * k + 2]; o[2] = row[2 * k + 4];
                    const bool keyed = c == 3 && trns.size() >= 6 && sample16(k) == (unsigned)(trns[0] << 8 | trns[1]) &&
                                       sample16(k + 1) == (unsigned)(trns[2] << 8 | trns[3]) && sample16(k + 2) == (unsigned)(trns[4] << 8 | trns[5]);
                    o[3] = c == 4 ? row[2 * k + 6] : (keyed ? 0 : 255);
                }
            }
        };
        std::vector<uint8_t> img;
        if (!interlace) {
            const size_t stride = ((size_t)w * bits + 7) / 8;
            if (!unfilter(raw.data(), raw.size(), h, stride, bpp, img)) return "corrupt image data";
            for (unsigned y = 0; y < h; ++y)
                for (unsigned x = 0; x < w; ++x) put(&img[y * stride], x, x, y);
        } else {  // Adam7: seven reduced images, one after the other
            static const unsigned x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
            size_t off = 0;
            for (int p = 0; p < 7; ++p) {
                const unsigned pw = (w + dx[p] - 1 - x0[p]) / dx[p], ph = (h + dy[p] - 1 - y0[p]) / dy[p];
                if (w <= x0[p] || h <= y0[p] || pw == 0 || ph == 0) continue;
                const size_t stride = ((size_t)pw * bits + 7) / 8;
                if (off > raw.size() || !unfilter(raw.data() + off, raw.size() - off, ph, stride, bpp, img)) return "corrupt image data";
                off += (size_t)ph * (stride + 1);
                for (unsigned y = 0; y < ph; ++y)
                    for (unsigned x = 0; x < pw; ++x) put(&img[y * stride], x, x0[p] + x * dx[p], y0[p] + y * dy[p]);
            }
        }
    } catch (const std::bad_alloc &) {
        rgba.clear();
        return "out of memory while decoding";
    }
    return "";
}

}  // namespace png_min
