// jpeg_decode.cpp -- JPEG (ITU-T T.81) decoder for textures: Huffman-coded baseline, extended-sequential and
// progressive DCT files with 8-bit samples, one (grey) or three (YCbCr or RGB) components.
//
// Upstream hands texture files to cv::imread(IMREAD_COLOR) (csrc/sceneloader.cpp:224), which for JPEG is
// libjpeg-turbo with its defaults.  The steps that decide the pixel values follow that library's default algorithms
// so that a texture decodes to the same bytes: the 13-bit fixed-point "slow integer" inverse DCT, the triangle-filter
// ("fancy") chroma upsampling for 2:1 ratios and replication for the others, and the 16-bit fixed-point
// YCbCr -> RGB tables.  Arithmetic-coded, lossless, hierarchical, 12-bit and four-component files are refused.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.h"

namespace rth {

namespace {

const uint8_t kZigzag[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};     // a run past the end lands on the last entry

struct HuffTable {
    bool defined = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    // canonical decoding (T.81 F.2.2.3): per code length the largest code and the index of its first value
    int32_t maxcode[18];
    int32_t valptr[17];
    uint16_t look[512];                 // 9-bit prefix -> (length << 8) | value, 0 when the code is longer
    void build() {
        int32_t code = 0, k = 0;
        std::memset(look, 0, sizeof(look));
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k - code;
            for (int i = 0; i < bits[len]; i++, k++, code++) {
                if (len <= 9 && k < 256) {
                    const int first = code << (9 - len);
                    for (int f = 0; f < (1 << (9 - len)); f++) look[(first + f) & 511] = uint16_t((len << 8) | vals[k]);
                }
            }
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int width = 0, height = 0;          // samples of this component that carry image data
    int blocks_w = 0, blocks_h = 0;     // allocated blocks: whole MCUs
    std::vector<int16_t> coef;          // blocks_w * blocks_h * 64, natural order inside a block
    std::vector<uint8_t> plane;         // blocks_w*8 by blocks_h*8 samples after the inverse DCT
    int dc_table = 0, ac_table = 0;     // of the current scan
    int last_dc = 0;
};

struct BitReader {
    const uint8_t* data;
    size_t size, pos;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    BitReader(const uint8_t* d, size_t s, size_t p) : data(d), size(s), pos(p) {}
    void fill() {
        while (n <= 24) {
            int byte = 0;
            if (!hit_marker && pos < size) {
                byte = data[pos];
                if (byte == 0xFF) {
                    size_t q = pos + 1;
                    while (q < size && data[q] == 0xFF) q++;           // fill bytes
                    if (q < size && data[q] == 0) { pos = q + 1; }      // stuffed zero: a data byte 0xFF
                    else { hit_marker = true; byte = 0; pos = q > 0 ? q - 1 : 0; }    // a marker: zeros from here on
                } else {
                    pos++;
                }
            }
            acc |= uint32_t(byte) << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return int(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    int bit() { return get(1); }
    // Align to the next marker and step over it when it is the expected restart marker.
    void restart() {
        acc = 0; n = 0;
        if (!hit_marker) {                                   // data left before the marker: skip it
            while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0 && data[pos + 1] != 0xFF)) pos++;
        }
        if (pos + 1 < size && data[pos] == 0xFF && data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7) pos += 2;
        hit_marker = false;
    }
};

inline int decode_symbol(BitReader& br, const HuffTable& t) {
    const int p = br.peek(16);
    const uint16_t e = t.look[p >> 7];
    if (e) { br.skip(e >> 8); return e & 0xFF; }
    for (int len = 10; len <= 16; len++) {
        const int32_t code = p >> (16 - len);
        if (code <= t.maxcode[len]) { br.skip(len); return t.vals[(code + t.valptr[len]) & 0xFF]; }
    }
    br.skip(16);
    return 0;                                                // a code that the table does not hold: as a zero symbol
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// The "slow integer" inverse DCT: 13 fractional bits, two extra bits kept between the column and the row pass.
// 64-bit intermediates, as the library has on LP64 systems: damaged files cannot overflow them.
const int kConstBits = 13, kPass1Bits = 2;
const int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
              F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;

inline int64_t descale(int64_t x, int n) { return (x + (int64_t(1) << (n - 1))) >> n; }

inline uint8_t range_limit(int64_t v) {                      // ten-bit wrap, then clamp around the level shift of 128
    const int64_t x = ((v & 1023) ^ 512) - 512 + 128;
    return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x));
}

void idct_block(const int16_t* coef, const uint16_t* quant, uint8_t* out, size_t stride) {
    int64_t ws[64];
    for (int c = 0; c < 8; c++) {
        int64_t in[8];
        for (int r = 0; r < 8; r++) in[r] = int64_t(int32_t(coef[r * 8 + c]) * int32_t(quant[r * 8 + c]));
        int64_t z2 = in[2], z3 = in[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = in[0]; z3 = in[4];
        int64_t tmp0 = (z2 + z3) * (1 << kConstBits), tmp1 = (z2 - z3) * (1 << kConstBits);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int sh = kConstBits - kPass1Bits;
        ws[0 * 8 + c] = descale(tmp10 + tmp3, sh); ws[7 * 8 + c] = descale(tmp10 - tmp3, sh);
        ws[1 * 8 + c] = descale(tmp11 + tmp2, sh); ws[6 * 8 + c] = descale(tmp11 - tmp2, sh);
        ws[2 * 8 + c] = descale(tmp12 + tmp1, sh); ws[5 * 8 + c] = descale(tmp12 - tmp1, sh);
        ws[3 * 8 + c] = descale(tmp13 + tmp0, sh); ws[4 * 8 + c] = descale(tmp13 - tmp0, sh);
    }
    for (int r = 0; r < 8; r++) {
        const int64_t* in = &ws[r * 8];
        int64_t z2 = in[2], z3 = in[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        int64_t tmp0 = (in[0] + in[4]) * (1 << kConstBits), tmp1 = (in[0] - in[4]) * (1 << kConstBits);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int sh = kConstBits + kPass1Bits + 3;
        uint8_t* o = out + size_t(r) * stride;
        o[0] = range_limit(descale(tmp10 + tmp3, sh)); o[7] = range_limit(descale(tmp10 - tmp3, sh));
        o[1] = range_limit(descale(tmp11 + tmp2, sh)); o[6] = range_limit(descale(tmp11 - tmp2, sh));
        o[2] = range_limit(descale(tmp12 + tmp1, sh)); o[5] = range_limit(descale(tmp12 - tmp1, sh));
        o[3] = range_limit(descale(tmp13 + tmp0, sh)); o[4] = range_limit(descale(tmp13 - tmp0, sh));
    }
}

struct Decoder {
    const std::vector<uint8_t>& file;
    std::string& error;
    int width = 0, height = 0, precision = 0;
    bool progressive = false, have_frame = false;
    bool saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    int hmax = 1, vmax = 1, mcus_w = 0, mcus_h = 0;
    int restart_interval = 0;
    std::vector<Component> comps;
    uint16_t quant[4][64];
    bool quant_defined[4] = {false, false, false, false};
    HuffTable dc_tables[4], ac_tables[4];

    Decoder(const std::vector<uint8_t>& f, std::string& e) : file(f), error(e) { std::memset(quant, 0, sizeof(quant)); }

    bool fail(const char* what) { error = std::string("JPEG: ") + what; return false; }

    bool read_dqt(size_t p, size_t end) {
        while (p < end) {
            const int pq = file[p] >> 4, tq = file[p] & 15;
            p++;
            if (tq > 3 || pq > 1) return fail("bad quantisation table header");
            if (p + size_t(64 * (pq + 1)) > end) return fail("truncated quantisation table");
            for (int i = 0; i < 64; i++) {
                const int v = pq ? ((file[p] << 8) | file[p + 1]) : file[p];
                p += size_t(pq + 1);
                quant[tq][kZigzag[i]] = uint16_t(v);
            }
            quant_defined[tq] = true;
        }
        return true;
    }

    bool read_dht(size_t p, size_t end) {
        while (p < end) {
            const int tc = file[p] >> 4, th = file[p] & 15;
            p++;
            if (tc > 1 || th > 3) return fail("bad Huffman table header");
            if (p + 16 > end) return fail("truncated Huffman table");
            HuffTable& t = tc ? ac_tables[th] : dc_tables[th];
            int total = 0;
            t.bits[0] = 0;
            for (int i = 1; i <= 16; i++) { t.bits[i] = file[p + size_t(i) - 1]; total += t.bits[i]; }
            p += 16;
            if (total > 256 || p + size_t(total) > end) return fail("bad Huffman table");
            std::memset(t.vals, 0, sizeof(t.vals));
            for (int i = 0; i < total; i++) t.vals[i] = file[p + size_t(i)];
            p += size_t(total);
            t.build();
        }
        return true;
    }

    bool read_sof(size_t p, size_t end) {
        if (have_frame) return fail("more than one frame");
        if (end - p < 6) return fail("truncated frame header");
        precision = file[p];
        height = (file[p + 1] << 8) | file[p + 2];
        width = (file[p + 3] << 8) | file[p + 4];
        const int n = file[p + 5];
        if (precision != 8) return fail("only 8-bit samples are decoded");
        if (width <= 0 || height <= 0) return fail("empty image");
        if (int64_t(width) * int64_t(height) > (int64_t(1) << 28)) return fail("image too large");
        if (n != 1 && n != 3) return fail("only one- and three-component files are decoded");
        if (end - p < size_t(6 + 3 * n)) return fail("truncated frame header");
        comps.resize(size_t(n));
        for (int i = 0; i < n; i++) {
            Component& c = comps[size_t(i)];
            c.id = file[p + 6 + size_t(3 * i)];
            c.h = file[p + 7 + size_t(3 * i)] >> 4;
            c.v = file[p + 7 + size_t(3 * i)] & 15;
            c.tq = file[p + 8 + size_t(3 * i)];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component in the frame header");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        if (n == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }       // a lone component is never interleaved
        mcus_w = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_h = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            c.width = (width * c.h + hmax - 1) / hmax;
            c.height = (height * c.v + vmax - 1) / vmax;
            c.blocks_w = mcus_w * c.h;
            c.blocks_h = mcus_h * c.v;
            c.coef.assign(size_t(c.blocks_w) * size_t(c.blocks_h) * 64, 0);
        }
        have_frame = true;
        return true;
    }

    // one block of a sequential scan: DC difference, then run/size coded AC coefficients
    void sequential_block(BitReader& br, Component& c, int16_t* blk) {
        const int s = decode_symbol(br, dc_tables[c.dc_table]) & 15;
        if (s) c.last_dc += extend(br.get(s), s);
        blk[0] = int16_t(c.last_dc);
        const HuffTable& ac = ac_tables[c.ac_table];
        for (int k = 1; k < 64; k++) {
            const int rs = decode_symbol(br, ac), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                blk[kZigzag[k]] = int16_t(extend(br.get(sz), sz));
            } else {
                if (r != 15) break;
                k += 15;
            }
        }
    }

    void dc_first(BitReader& br, Component& c, int16_t* blk, int al) {
        const int s = decode_symbol(br, dc_tables[c.dc_table]) & 15;
        if (s) c.last_dc += extend(br.get(s), s);
        blk[0] = int16_t(c.last_dc * (1 << al));
    }

    void ac_first(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al, uint32_t& eobrun) {
        if (eobrun > 0) { eobrun--; return; }
        const HuffTable& ac = ac_tables[c.ac_table];
        for (int k = ss; k <= se; k++) {
            const int rs = decode_symbol(br, ac), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                blk[kZigzag[k]] = int16_t(extend(br.get(sz), sz) * (1 << al));
            } else if (r == 15) {
                k += 15;
            } else {
                eobrun = uint32_t(1) << r;
                if (r) eobrun += uint32_t(br.get(r));
                eobrun--;
                break;
            }
        }
    }

    void ac_refine(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al, uint32_t& eobrun) {
        const int p1 = 1 << al, m1 = -(1 << al);
        const HuffTable& ac = ac_tables[c.ac_table];
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; k++) {
                const int rs = decode_symbol(br, ac);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = br.bit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun = uint32_t(1) << r;
                    if (r) eobrun += uint32_t(br.get(r));
                    break;
                }
                do {
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0) {
                        if (br.bit() && (v & p1) == 0) v = int16_t(v + (v >= 0 ? p1 : m1));
                    } else if (--r < 0) {
                        break;
                    }
                    k++;
                } while (k <= se);
                if (s && k < 64) blk[kZigzag[k]] = int16_t(s);
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0 && br.bit() && (v & p1) == 0) v = int16_t(v + (v >= 0 ? p1 : m1));
            }
            eobrun--;
        }
    }

    // A scan: header at p (after the length field); returns the position after its entropy-coded data.
    bool read_scan(size_t p, size_t end, size_t& after) {
        if (!have_frame) return fail("scan before the frame header");
        if (end - p < 1) return fail("truncated scan header");
        const int ns = file[p];
        if (ns < 1 || ns > int(comps.size()) || end - p < size_t(4 + 2 * ns)) return fail("bad scan header");
        std::vector<Component*> sc;
        for (int i = 0; i < ns; i++) {
            const int id = file[p + 1 + size_t(2 * i)], tables = file[p + 2 + size_t(2 * i)];
            Component* found = nullptr;
            for (Component& c : comps) if (c.id == id) found = &c;
            if (!found) return fail("scan names an unknown component");
            found->dc_table = tables >> 4;
            found->ac_table = tables & 15;
            if (found->dc_table > 3 || found->ac_table > 3) return fail("bad table selector");
            sc.push_back(found);
        }
        const int ss = file[p + 1 + size_t(2 * ns)], se = file[p + 2 + size_t(2 * ns)];
        const int ah = file[p + 3 + size_t(2 * ns)] >> 4, al = file[p + 3 + size_t(2 * ns)] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            // sequential files with other values are decoded as if they were the standard ones, as the library does
        }
        const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
        for (Component* c : sc) {
            if (need_dc && !(progressive && ah != 0) && !dc_tables[c->dc_table].defined) return fail("scan uses an undefined DC table");
            if (need_ac && !ac_tables[c->ac_table].defined) return fail("scan uses an undefined AC table");
            c->last_dc = 0;
        }
        BitReader br(file.data(), file.size(), end);
        uint32_t eobrun = 0;
        int until_restart = restart_interval;
        auto do_block = [&](Component& c, int bx, int by) {
            int16_t* blk = &c.coef[(size_t(by) * size_t(c.blocks_w) + size_t(bx)) * 64];
            if (!progressive) sequential_block(br, c, blk);
            else if (ss == 0) { if (ah == 0) dc_first(br, c, blk, al); else if (br.bit()) blk[0] = int16_t(blk[0] | (1 << al)); }
            else if (ah == 0) ac_first(br, c, blk, ss, se, al, eobrun);
            else ac_refine(br, c, blk, ss, se, al, eobrun);
        };
        auto after_mcu = [&](bool last) {
            if (restart_interval && !last && --until_restart == 0) {
                br.restart();
                for (Component* c : sc) c->last_dc = 0;
                eobrun = 0;
                until_restart = restart_interval;
            }
        };
        if (ns == 1) {                                       // not interleaved: the component's own block grid
            Component& c = *sc[0];
            const int bw = (c.width + 7) / 8, bh = (c.height + 7) / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    do_block(c, bx, by);
                    after_mcu(by == bh - 1 && bx == bw - 1);
                }
        } else {
            for (int my = 0; my < mcus_h; my++)
                for (int mx = 0; mx < mcus_w; mx++) {
                    for (Component* c : sc)
                        for (int v = 0; v < c->v; v++)
                            for (int h = 0; h < c->h; h++) do_block(*c, mx * c->h + h, my * c->v + v);
                    after_mcu(my == mcus_h - 1 && mx == mcus_w - 1);
                }
        }
        // the next marker: where the reader met it, or the first one after the data it consumed
        size_t q = br.pos;
        while (q + 1 < file.size() && !(file[q] == 0xFF && file[q + 1] != 0 && file[q + 1] != 0xFF && !(file[q + 1] >= 0xD0 && file[q + 1] <= 0xD7))) q++;
        after = q;
        return true;
    }

    // ---- after the last scan ----

    void inverse_transform() {
        for (Component& c : comps) {
            const size_t stride = size_t(c.blocks_w) * 8;
            c.plane.assign(stride * size_t(c.blocks_h) * 8, 0);
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++)
                    idct_block(&c.coef[(size_t(by) * size_t(c.blocks_w) + size_t(bx)) * 64], quant[c.tq], &c.plane[size_t(by) * 8 * stride + size_t(bx) * 8], stride);
        }
    }

    // One component at full resolution (width x height).  2:1 ratios use the triangle filter: each output sample is
    // 3/4 of the nearer and 1/4 of the further input sample, with rounding constants that alternate between 1 and 2
    // (in quarters) or 8 and 7 (in sixteenths); every other ratio replicates.
    void upsample(const Component& c, std::vector<uint8_t>& full) const {
        const size_t stride = size_t(c.blocks_w) * 8;
        const int hr = hmax / c.h, vr = vmax / c.v;
        const bool exact = (hmax % c.h == 0) && (vmax % c.v == 0);
        full.assign(size_t(width) * size_t(height), 0);
        const int dw = c.width, dh = c.height;
        auto row_ptr = [&](int r) { return &c.plane[size_t(r < 0 ? 0 : (r >= dh ? dh - 1 : r)) * stride]; };
        if (exact && hr == 1 && vr == 1) {
            for (int y = 0; y < height; y++) std::memcpy(&full[size_t(y) * size_t(width)], row_ptr(y), size_t(width));
            return;
        }
        std::vector<uint8_t> line(static_cast<size_t>(dw) * 2 + 2, 0);
        if (exact && hr == 2 && vr == 1 && dw > 2) {
            for (int y = 0; y < height; y++) {
                const uint8_t* in = row_ptr(y);
                line[0] = in[0];
                line[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < dw - 1; i++) {
                    line[size_t(2 * i)] = uint8_t((in[i] * 3 + in[i - 1] + 1) >> 2);
                    line[size_t(2 * i + 1)] = uint8_t((in[i] * 3 + in[i + 1] + 2) >> 2);
                }
                line[size_t(2 * dw - 2)] = uint8_t((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
                line[size_t(2 * dw - 1)] = in[dw - 1];
                std::memcpy(&full[size_t(y) * size_t(width)], line.data(), size_t(width));
            }
            return;
        }
        if (exact && hr == 2 && vr == 2 && dw > 2) {
            std::vector<int> sum(static_cast<size_t>(dw), 0);
            for (int y = 0; y < height; y++) {
                const uint8_t* near = row_ptr(y >> 1);
                const uint8_t* far = row_ptr((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1);
                for (int i = 0; i < dw; i++) sum[size_t(i)] = near[i] * 3 + far[i];
                line[0] = uint8_t((sum[0] * 4 + 8) >> 4);
                line[1] = uint8_t((sum[0] * 3 + sum[1] + 7) >> 4);
                for (int i = 1; i < dw - 1; i++) {
                    line[size_t(2 * i)] = uint8_t((sum[size_t(i)] * 3 + sum[size_t(i - 1)] + 8) >> 4);
                    line[size_t(2 * i + 1)] = uint8_t((sum[size_t(i)] * 3 + sum[size_t(i + 1)] + 7) >> 4);
                }
                line[size_t(2 * dw - 2)] = uint8_t((sum[size_t(dw - 1)] * 3 + sum[size_t(dw - 2)] + 8) >> 4);
                line[size_t(2 * dw - 1)] = uint8_t((sum[size_t(dw - 1)] * 4 + 7) >> 4);
                std::memcpy(&full[size_t(y) * size_t(width)], line.data(), size_t(width));
            }
            return;
        }
        if (exact && hr == 1 && vr == 2) {
            for (int y = 0; y < height; y++) {
                const uint8_t* near = row_ptr(y >> 1);
                const uint8_t* far = row_ptr((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1);
                const int bias = (y & 1) ? 2 : 1;
                uint8_t* o = &full[size_t(y) * size_t(width)];
                for (int x = 0; x < width; x++) o[x] = uint8_t((near[x] * 3 + far[x] + bias) >> 2);
            }
            return;
        }
        // replication (whole ratios), or nearest sample for the ratios the library itself would refuse
        for (int y = 0; y < height; y++) {
            const uint8_t* in = row_ptr(exact ? y / vr : int((int64_t(y) * c.v) / vmax));
            uint8_t* o = &full[size_t(y) * size_t(width)];
            for (int x = 0; x < width; x++) {
                int sx = exact ? x / hr : int((int64_t(x) * c.h) / hmax);
                o[x] = in[sx >= dw ? dw - 1 : sx];
            }
        }
    }

    bool finish(Image8& out) {
        for (const Component& c : comps) if (!quant_defined[c.tq]) return fail("component uses an undefined quantisation table");
        inverse_transform();
        out.width = width; out.height = height;
        out.rgb.resize(size_t(width) * size_t(height) * 3);
        std::vector<uint8_t> plane[3];
        for (size_t i = 0; i < comps.size(); i++) upsample(comps[i], plane[i]);
        const size_t n = size_t(width) * size_t(height);
        if (comps.size() == 1) {
            for (size_t i = 0; i < n; i++) out.rgb[3 * i] = out.rgb[3 * i + 1] = out.rgb[3 * i + 2] = plane[0][i];
            return true;
        }
        bool ycc = true;                                     // the library's colour-space guess for three components
        if (saw_jfif) ycc = true;
        else if (saw_adobe) ycc = adobe_transform != 0;
        else if (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
        if (!ycc) {
            for (size_t i = 0; i < n; i++) { out.rgb[3 * i] = plane[0][i]; out.rgb[3 * i + 1] = plane[1][i]; out.rgb[3 * i + 2] = plane[2][i]; }
            return true;
        }
        // 16-bit fixed point: R = Y + 1.402 Cr, G = Y - 0.34414 Cb - 0.71414 Cr, B = Y + 1.772 Cb, the chroma terms
        // tabulated per sample value with the rounding of the library's tables
        int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        const int32_t half = 1 << 15;
        for (int i = 0; i < 256; i++) {
            const int32_t x = i - 128;
            cr_r[i] = (91881 * x + half) >> 16;
            cb_b[i] = (116130 * x + half) >> 16;
            cr_g[i] = -46802 * x;
            cb_g[i] = -22554 * x + half;
        }
        auto clamp8 = [](int32_t v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (size_t i = 0; i < n; i++) {
            const int y = plane[0][i], cb = plane[1][i], cr = plane[2][i];
            out.rgb[3 * i] = clamp8(y + cr_r[cr]);
            out.rgb[3 * i + 1] = clamp8(y + ((cb_g[cb] + cr_g[cr]) >> 16));
            out.rgb[3 * i + 2] = clamp8(y + cb_b[cb]);
        }
        return true;
    }

    bool run(Image8& out) {
        size_t p = 2;
        bool any_scan = false;
        while (p + 4 <= file.size()) {
            if (file[p] != 0xFF) { p++; continue; }
            const int m = file[p + 1];
            if (m == 0xFF) { p++; continue; }
            if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            if (m == 0xD9) break;
            const size_t len = (size_t(file[p + 2]) << 8) | file[p + 3];
            if (len < 2 || p + 2 + len > file.size()) return fail("truncated segment");
            const size_t body = p + 4, end = p + 2 + len;
            size_t next = end;
            switch (m) {
            case 0xC0: case 0xC1: case 0xC2:
                progressive = (m == 0xC2);
                if (!read_sof(body, end)) return false;
                break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("lossless, hierarchical and arithmetic-coded files are not decoded");
            case 0xC4: if (!read_dht(body, end)) return false; break;
            case 0xDB: if (!read_dqt(body, end)) return false; break;
            case 0xDD:
                if (len < 4) return fail("bad restart interval segment");
                restart_interval = (file[body] << 8) | file[body + 1];
                break;
            case 0xE0:
                if (len >= 7 && !std::memcmp(&file[body], "JFIF", 5)) saw_jfif = true;
                break;
            case 0xEE:
                if (len >= 14 && !std::memcmp(&file[body], "Adobe", 5)) { saw_adobe = true; adobe_transform = file[body + 11]; }
                break;
            case 0xDA:
                if (!read_scan(body, end, next)) return false;
                any_scan = true;
                break;
            default: break;
            }
            p = next;
        }
        if (!have_frame || !any_scan) return fail("no image data");
        return finish(out);
    }
};

}  // namespace

bool decode_jpeg(const std::vector<uint8_t>& file, Image8& out, std::string& error) {
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) { error = "JPEG: missing start-of-image marker"; return false; }
    Decoder d(file, error);
    return d.run(out);
}

}  // namespace rth
