// image_io.cpp -- see image_io.h.  JPEG lives in jpeg_decode.cpp.  PNG per the PNG specification (ISO/IEC 15948): chunks with CRC-32, one zlib
// stream over the IDAT chunks, per-row filters 0-4, Adam7 interlacing.
#include "image_io.h"

#include "file_util.h"

#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace rth {

namespace {

const unsigned char kPngSignature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t>& file, const char type[4], const uint8_t* data, size_t n) {
    put_be32(file, uint32_t(n));
    const size_t start = file.size();
    file.insert(file.end(), type, type + 4);
    if (n) file.insert(file.end(), data, data + n);
    put_be32(file, uint32_t(crc32(0L, file.data() + start, uInt(n + 4))));
}

bool read_file(const std::string& path, std::vector<uint8_t>& bytes) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f || !is_plain_file(path)) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0, std::ios::beg);
    bytes.resize(size_t(n));
    if (n) f.read(reinterpret_cast<char*>(bytes.data()), n);
    return bool(f) || f.eof();
}

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

bool decode_png(const std::vector<uint8_t>& file, Image8& out, std::string& error) {
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool seen_end = false;
    while (pos + 12 <= file.size() && !seen_end) {
        const uint32_t n = be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if (pos + 12 + size_t(n) > file.size()) { error = "PNG: truncated chunk"; return false; }
        const uint8_t* data = &file[pos + 8];
        if (crc32(0L, &file[pos + 4], uInt(n + 4)) != be32(&file[pos + 8 + n])) { error = "PNG: bad chunk CRC"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (n != 13) { error = "PNG: bad IHDR"; return false; }
            width = be32(data); height = be32(data + 4);
            depth = data[8]; colour = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { error = "PNG: unknown compression or filter method"; return false; }
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + n);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + n);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + size_t(n);
    }
    if (colour < 0 || width == 0 || height == 0 || width > 65535u || height > 65535u) { error = "PNG: missing or implausible IHDR"; return false; }
    if (interlace > 1) { error = "PNG: unknown interlace method"; return false; }
    int channels;
    switch (colour) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: error = "PNG: unknown colour type"; return false;
    }
    const bool depth_ok = (colour == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (colour == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((colour == 2 || colour == 4 || colour == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) { error = "PNG: bit depth not allowed for the colour type"; return false; }
    if (colour == 3 && palette.size() < 3) { error = "PNG: palette image without PLTE"; return false; }
    const size_t bits_per_pixel = size_t(channels) * size_t(depth);
    const size_t bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;      // filter distance in bytes
    // the passes of the file: the whole image, or the seven Adam7 sub-images (x0, y0, dx, dy), each filtered on its own
    struct Pass { size_t x0, y0, dx, dy, w, h, stride; };
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    std::vector<Pass> passes;
    size_t total = 0;
    for (int k = 0; k < (interlace ? 7 : 1); k++) {
        Pass ps;
        if (interlace) { ps.x0 = size_t(adam7[k][0]); ps.y0 = size_t(adam7[k][1]); ps.dx = size_t(adam7[k][2]); ps.dy = size_t(adam7[k][3]); }
        else { ps.x0 = ps.y0 = 0; ps.dx = ps.dy = 1; }
        ps.w = (size_t(width) + ps.dx - 1 - ps.x0) / ps.dx;
        ps.h = (size_t(height) + ps.dy - 1 - ps.y0) / ps.dy;
        if (size_t(width) <= ps.x0 || size_t(height) <= ps.y0) ps.w = ps.h = 0;
        ps.stride = (ps.w * bits_per_pixel + 7) / 8;
        if (ps.w == 0 || ps.h == 0) continue;                            // an empty pass has no bytes at all
        total += (ps.stride + 1) * ps.h;
        passes.push_back(ps);
    }
    std::vector<uint8_t> raw(total);
    uLongf raw_len = uLongf(raw.size());
    if (idat.empty() || uncompress(raw.data(), &raw_len, idat.data(), uLong(idat.size())) != Z_OK || raw_len != raw.size()) {
        error = "PNG: image data does not inflate to the declared size";
        return false;
    }
    out.width = int(width); out.height = int(height);
    out.rgb.resize(size_t(width) * height * 3);
    size_t base = 0;
    for (const Pass& ps : passes) {
        // undo the row filters in place
        std::vector<uint8_t> zero(ps.stride, 0);
        for (size_t y = 0; y < ps.h; y++) {
            uint8_t* row = &raw[base + y * (ps.stride + 1) + 1];
            const uint8_t* up = y ? &raw[base + (y - 1) * (ps.stride + 1) + 1] : zero.data();
            const int filter = raw[base + y * (ps.stride + 1)];
            for (size_t i = 0; i < ps.stride; i++) {
                const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
                int pred;
                switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = paeth(a, b, c); break;
                default: error = "PNG: unknown row filter"; return false;
                }
                row[i] = uint8_t(row[i] + pred);
            }
        }
        for (size_t y = 0; y < ps.h; y++) {
            const uint8_t* row = &raw[base + y * (ps.stride + 1) + 1];
            uint8_t* dst = &out.rgb[(ps.y0 + y * ps.dy) * size_t(width) * 3];
            for (size_t x = 0; x < ps.w; x++) {
                uint8_t r, g, b;
                if (depth < 8) {                     // grey or palette index packed high bit first
                    const size_t bit = x * size_t(depth);
                    const int v = (row[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1);
                    if (colour == 3) {
                        const size_t k = size_t(v) * 3;
                        if (k + 2 < palette.size()) { r = palette[k]; g = palette[k + 1]; b = palette[k + 2]; }
                        else { r = g = b = 0; }
                    } else {
                        r = g = b = uint8_t(v * 255 / ((1 << depth) - 1));
                    }
                } else {
                    const size_t step = size_t(depth) / 8;            // 16-bit samples: the high byte is kept
                    const uint8_t* px = row + x * size_t(channels) * step;
                    if (colour == 3) {
                        const size_t k = size_t(px[0]) * 3;
                        if (k + 2 < palette.size()) { r = palette[k]; g = palette[k + 1]; b = palette[k + 2]; }
                        else { r = g = b = 0; }
                    } else if (colour == 0 || colour == 4) {
                        r = g = b = px[0];
                    } else {
                        r = px[0]; g = px[step]; b = px[2 * step];
                    }
                }
                uint8_t* o = dst + (ps.x0 + x * ps.dx) * 3;
                o[0] = r; o[1] = g; o[2] = b;
            }
        }
        base += (ps.stride + 1) * ps.h;
    }
    return true;
}

// "P6" / "P5": header tokens separated by whitespace, '#' comments, one whitespace byte, then samples
bool decode_pnm(const std::vector<uint8_t>& file, Image8& out, std::string& error) {
    const bool grey = file[1] == '5';
    size_t pos = 2;
    long fields[3];
    for (int k = 0; k < 3; k++) {
        for (;;) {
            while (pos < file.size() && std::isspace(file[pos])) pos++;
            if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
            break;
        }
        long v = 0;
        size_t digits = 0;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { v = v * 10 + (file[pos] - '0'); pos++; digits++; if (v > 1000000) break; }
        if (!digits) { error = "PNM: bad header"; return false; }
        fields[k] = v;
    }
    pos++;                                        // the single whitespace byte after maxval
    const long w = fields[0], h = fields[1], maxval = fields[2];
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || maxval <= 0 || maxval > 255) { error = "PNM: unsupported size or maxval"; return false; }
    const size_t need = size_t(w) * size_t(h) * (grey ? 1 : 3);
    if (pos + need > file.size()) { error = "PNM: truncated"; return false; }
    out.width = int(w); out.height = int(h);
    out.rgb.resize(size_t(w) * size_t(h) * 3);
    for (size_t i = 0; i < size_t(w) * size_t(h); i++) {
        for (int c = 0; c < 3; c++) {
            const int v = file[pos + (grey ? i : 3 * i + size_t(c))];
            out.rgb[3 * i + size_t(c)] = uint8_t(maxval == 255 ? v : v * 255 / maxval);
        }
    }
    return true;
}

}  // namespace

bool read_image(const std::string& path, Image8& out, std::string& error) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { error = "cannot read " + path; return false; }
    if (file.size() >= 8 && !std::memcmp(file.data(), kPngSignature, 8)) return decode_png(file, out, error);
    if (file.size() >= 8 && file[0] == 'P' && (file[1] == '6' || file[1] == '5')) return decode_pnm(file, out, error);
    if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8) return decode_jpeg(file, out, error);
    error = "unknown image format: " + path;
    return false;
}

bool write_png_rgb8(const std::string& path, int width, int height, const uint8_t* rgb, std::string& error) {
    if (width <= 0 || height <= 0) { error = "write_png_rgb8: empty image"; return false; }
    const size_t stride = size_t(width) * 3;
    std::vector<uint8_t> raw((stride + 1) * size_t(height));
    for (int y = 0; y < height; y++) {
        raw[size_t(y) * (stride + 1)] = 0;                                     // filter "none"
        std::memcpy(&raw[size_t(y) * (stride + 1) + 1], rgb + size_t(y) * stride, stride);
    }
    // Fastest level, run-length matching only: a path-traced frame is noisy, so longer match searches buy nothing (a
    // 2000x1200 frame: 7.0 MB in 0.10 s against 6.9 MB in 0.39 s at level 6), and the viewer waits for this file after
    // every frame.  (OpenCV's imwrite, which upstream calls at csrc/worker.cpp:40, defaults to the same kind of setting.)
    uLongf zlen = compressBound(uLong(raw.size())) + 64;
    std::vector<uint8_t> z(zlen);
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 1, Z_DEFLATED, 15, 8, Z_RLE) != Z_OK) { error = "write_png_rgb8: deflate failed"; return false; }
    zs.next_in = raw.data(); zs.avail_in = uInt(raw.size());
    zs.next_out = z.data(); zs.avail_out = uInt(z.size());
    const int zrc = deflate(&zs, Z_FINISH);
    zlen = zs.total_out;
    deflateEnd(&zs);
    if (zrc != Z_STREAM_END) { error = "write_png_rgb8: deflate failed"; return false; }
    std::vector<uint8_t> file(kPngSignature, kPngSignature + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, uint32_t(width)); put_be32(ihdr, uint32_t(height));
    const uint8_t tail[5] = {8, 2, 0, 0, 0};                                   // 8 bits, truecolour, deflate, adaptive filters, not interlaced
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(file, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(file, "IDAT", z.data(), size_t(zlen));
    put_chunk(file, "IEND", nullptr, 0);
    // write to a temporary name and rename, so a reader never sees half a file
    const std::string tmp = path + ".tmp";
    {
        std::ofstream f(tmp.c_str(), std::ios::binary | std::ios::trunc);
        if (!f) { error = "cannot write " + tmp; return false; }
        f.write(reinterpret_cast<const char*>(file.data()), std::streamsize(file.size()));
        if (!f) { error = "short write to " + tmp; return false; }
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { error = "cannot rename " + tmp + " to " + path; return false; }
    return true;
}

uint8_t float_to_u8(float v) {
    const float s = v * 255.0f;
    if (!(s == s)) return 0;
    if (s <= 0.0f) return 0;
    if (s >= 255.0f) return 255;
    return uint8_t(std::nearbyintf(s));           // default rounding mode: to nearest, ties to even
}

void frame_to_rgb8(const float* f, int width, int height, std::vector<uint8_t>& out) {
    out.resize(size_t(width) * size_t(height) * 3);
    for (int y = 0; y < height; y++) {
        const float* src = f + size_t(height - 1 - y) * size_t(width) * 3;
        uint8_t* dst = &out[size_t(y) * size_t(width) * 3];
        for (int i = 0; i < width * 3; i++) dst[i] = float_to_u8(src[i]);
    }
}

}  // namespace rth
