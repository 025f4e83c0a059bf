// jpeg.cpp — JPEG (ITU-T T.81: sequential and progressive DCT, Huffman, 8 bit) -> RGBA8 for the glTF / OBJ importers.
//
// The reference reads images through the `image` crate inside the un-vendored crate l3d 0.3 (crates/rfw-scene/src/loaders/gltf.rs:26-90),
// and its own sample asset assets/models/CesiumMan/CesiumMan.jpg — the texture of the skinned model examples/animated loads
// (examples/animated/src/main.rs:80) — is a baseline 4:4:4 JPEG with restart intervals.  This decoder follows the standard: marker
// segments (DQT, DHT, SOF0/SOF1, DRI, SOS, APP14), interleaved and single-component scans, restart markers, any sampling factors
// (2x1 and 2x2 chroma through the usual triangle filter, other ratios by replication), the JFIF YCbCr -> RGB conversion in the 16-bit fixed
// point every libjpeg descendant uses.  Progressive files (SOF2: spectral selection and successive approximation, T.81 annex G) are decoded
// into the same coefficient store, scan by scan; arithmetic-coded, lossless and 12-bit files are refused (the material then stays
// untextured, like any image the importer cannot read).  The inverse DCT is evaluated in double precision and rounded once, so samples can differ by one
// level from decoders with an integer IDCT; tests/test_gltf.py holds it against Pillow with that tolerance.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rfw {

namespace {

struct Huff {
    bool present = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int32_t mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    void build()
    {
        int32_t code = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            k += bits[l];
            code += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0; // allocated blocks (whole MCUs)
    int width = 0, height = 0;      // samples that carry image data: ceil(image * h / hmax)
    std::vector<uint8_t> samples;   // blocks_w * 8 per row
    std::vector<int32_t> coef;      // blocks_w * blocks_h blocks of 64 quantised coefficients, natural order (dequantised at the end)
    int dc_pred = 0;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    int bit()
    {
        if (n == 0) {
            uint8_t b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) p++; // stuffed zero
                    else { hit_marker = true; p--; b = 0; } // a marker: feed zeros (T.81 F.2.2.5), the caller re-synchronises
                }
            }
            acc = b;
            n = 8;
        }
        n--;
        return (int)((acc >> n) & 1u);
    }
    int receive(int s)
    {
        int v = 0;
        for (int i = 0; i < s; i++) v = (v << 1) | bit();
        return v;
    }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

bool decode_symbol(BitReader& br, const Huff& h, int& out)
{
    int32_t code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
            out = h.vals[h.valptr[l] + (code - h.mincode[l])];
            return true;
        }
    }
    return false;
}

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Idct {
    double c[8][8]; // c[x][u] = 0.5 * C(u) * cos((2x+1) u pi / 16)
    Idct()
    {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) c[x][u] = 0.5 * (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    }
    void run(const int32_t* coef, uint8_t* out, int stride) const
    {
        double tmp[64];
        for (int v = 0; v < 8; v++) // rows: over u
            for (int x = 0; x < 8; x++) {
                double s = 0.0;
                for (int u = 0; u < 8; u++) s += c[x][u] * (double)coef[v * 8 + u];
                tmp[v * 8 + x] = s;
            }
        for (int x = 0; x < 8; x++)
            for (int y = 0; y < 8; y++) {
                double s = 0.0;
                for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
                const long r = std::lround(s) + 128;
                out[y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
    }
};

} // namespace

bool decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err)
{
    auto fail = [&](const char* m) { err = m; return false; };
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("jpeg: no SOI marker");
    uint16_t qt[4][64] = {};
    bool qt_present[4] = {};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
    int adobe_transform = -1;
    bool have_frame = false, decoded_any = false, progressive = false;
    static const Idct idct;
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (data[pos] != 0xFF) return fail("jpeg: marker expected");
        while (pos < size && data[pos] == 0xFF) pos++; // fill bytes
        if (pos >= size) break;
        const uint8_t m = data[pos++];
        if (m == 0xD9) break;                  // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; // TEM / stray RSTn: no payload
        if (pos + 2 > size) return fail("jpeg: truncated segment");
        const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
        if (len < 2 || pos + len > size) return fail("jpeg: truncated segment");
        const uint8_t* s = data + pos + 2;
        const size_t n = len - 2;
        pos += len;
        if (m == 0xDB) { // DQT
            size_t o = 0;
            while (o < n) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                o++;
                if (tq > 3 || pq > 1 || o + (size_t)(pq ? 128 : 64) > n) return fail("jpeg: bad quantisation table");
                for (int i = 0; i < 64; i++) {
                    qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[o] << 8) | s[o + 1]) : s[o];
                    o += pq ? 2 : 1;
                }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) { // DHT
            size_t o = 0;
            while (o < n) {
                if (o + 17 > n) return fail("jpeg: bad Huffman table");
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return fail("jpeg: bad Huffman table id");
                Huff& h = tc ? ac[th] : dc[th];
                size_t total = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[o + (size_t)l]; total += h.bits[l]; }
                o += 17;
                if (total > 256 || o + total > n) return fail("jpeg: bad Huffman table");
                std::memcpy(h.vals, s + o, total);
                o += total;
                h.build();
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { // SOF0 / SOF1: sequential DCT; SOF2: progressive DCT; Huffman
            progressive = m == 0xC2;
            if (have_frame) return fail("jpeg: more than one frame");
            if (n < 6 || s[0] != 8) return fail("jpeg: only 8-bit samples are supported");
            H = (s[1] << 8) | s[2];
            W = (s[3] << 8) | s[4];
            const int nc = s[5];
            if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return fail("jpeg: unsupported frame header");
            if ((uint64_t)W * (uint64_t)H > (1ull << 26)) return fail("jpeg: image too large"); // 64 M pixels: the coefficient store is 12 bytes per pixel
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                Component& c = comps[(size_t)i];
                c.id = s[6 + 3 * i];
                c.h = s[7 + 3 * i] >> 4;
                c.v = s[7 + 3 * i] & 15;
                c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("jpeg: bad component");
                hmax = c.h > hmax ? c.h : hmax;
                vmax = c.v > vmax ? c.v : vmax;
            }
            const int mcus_x = (W + 8 * hmax - 1) / (8 * hmax), mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
            for (Component& c : comps) {
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.width = (W * c.h + hmax - 1) / hmax;
                c.height = (H * c.v + vmax - 1) / vmax;
                c.samples.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 128);
                c.coef.assign((size_t)c.blocks_w * (size_t)c.blocks_h * 64, 0);
            }
            have_frame = true;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return fail("jpeg: unsupported coding process (lossless, hierarchical or arithmetic)");
        } else if (m == 0xDD) { // DRI
            if (n < 2) return fail("jpeg: bad DRI");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE) { // APP14 "Adobe": colour transform flag
            if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        } else if (m == 0xDA) { // SOS + entropy-coded segment
            if (!have_frame) return fail("jpeg: scan before the frame header");
            if (n < 1) return fail("jpeg: bad scan header");
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail("jpeg: bad scan header");
            std::vector<Component*> sc;
            for (int i = 0; i < ns; i++) {
                Component* c = nullptr;
                for (Component& k : comps)
                    if (k.id == s[1 + 2 * i]) c = &k;
                if (!c) return fail("jpeg: scan names an unknown component");
                c->td = s[2 + 2 * i] >> 4;
                c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !qt_present[c->tq]) return fail("jpeg: scan uses a missing table");
                sc.push_back(c);
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (!progressive) {
                if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail("jpeg: not a sequential scan");
            } else {
                if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1))
                    return fail("jpeg: bad progressive scan parameters");
            }
            for (Component* c : sc) {
                const bool need_dc = !progressive || Ss == 0, need_ac = !progressive || Ss > 0;
                if ((need_dc && Ah == 0 && !dc[c->td].present) || (need_ac && !ac[c->ta].present)) return fail("jpeg: scan uses a missing table");
            }
            BitReader br{data + pos, data + size};
            for (Component* c : sc) c->dc_pred = 0;
            int eobrun = 0;
            // units of the scan: MCUs (interleaved) or the component's own blocks that carry image data (single component, T.81 A.2.2)
            const int mcus_x = (W + 8 * hmax - 1) / (8 * hmax), mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
            const int ux = ns > 1 ? mcus_x : (sc[0]->width + 7) / 8, uy = ns > 1 ? mcus_y : (sc[0]->height + 7) / 8;
            int until_restart = restart_interval, expect_rst = 0;
            for (int my = 0; my < uy; my++)
                for (int mx = 0; mx < ux; mx++) {
                    if (restart_interval && until_restart == 0) {
                        // re-synchronise on RSTn: drop the bits of the current byte, find the marker
                        const uint8_t* q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        if (q + 1 >= br.end) return fail("jpeg: restart marker missing");
                        if ((q[1] & 7) != expect_rst) return fail("jpeg: restart markers out of sequence");
                        expect_rst = (expect_rst + 1) & 7;
                        br.p = q + 2;
                        br.reset();
                        for (Component* c : sc) c->dc_pred = 0;
                        eobrun = 0;
                        until_restart = restart_interval;
                    }
                    for (Component* c : sc) {
                        const int bw = ns > 1 ? c->h : 1, bh = ns > 1 ? c->v : 1;
                        for (int by = 0; by < bh; by++)
                            for (int bx = 0; bx < bw; bx++) {
                                const int gx = mx * bw + bx, gy = my * bh + by;
                                int32_t scratch[64];
                                const bool inside = gx < c->blocks_w && gy < c->blocks_h;
                                int32_t* coef = inside ? c->coef.data() + ((size_t)gy * (size_t)c->blocks_w + (size_t)gx) * 64 : scratch;
                                if (!inside) std::memset(scratch, 0, sizeof(scratch));
                                if (!progressive) { // T.81 F.2.2
                                    int t = 0;
                                    if (!decode_symbol(br, dc[c->td], t) || t > 11) return fail("jpeg: bad DC code");
                                    c->dc_pred += extend(br.receive(t), t);
                                    coef[0] = c->dc_pred;
                                    for (int k = 1; k < 64;) {
                                        int rs = 0;
                                        if (!decode_symbol(br, ac[c->ta], rs)) return fail("jpeg: bad AC code");
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r == 15) { k += 16; continue; }
                                            break; // EOB
                                        }
                                        k += r;
                                        if (k > 63) return fail("jpeg: AC run past the block");
                                        coef[kZigzag[k]] = extend(br.receive(sz), sz);
                                        k++;
                                    }
                                } else if (Ss == 0) { // DC scan (G.1.2.1): first pass = the sequential DC code, scaled; refinement = one bit
                                    if (Ah == 0) {
                                        int t = 0;
                                        if (!decode_symbol(br, dc[c->td], t) || t > 11) return fail("jpeg: bad DC code");
                                        c->dc_pred += extend(br.receive(t), t);
                                        coef[0] = c->dc_pred * (1 << Al);
                                    } else if (br.bit()) {
                                        coef[0] |= 1 << Al;
                                    }
                                } else if (Ah == 0) { // AC scan, first pass (G.1.2.2): runs of zeros, end-of-band runs over several blocks
                                    if (eobrun > 0) { eobrun--; continue; }
                                    for (int k = Ss; k <= Se;) {
                                        int rs = 0;
                                        if (!decode_symbol(br, ac[c->ta], rs)) return fail("jpeg: bad AC code");
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r < 15) { eobrun = (1 << r) - 1 + (r ? br.receive(r) : 0); break; }
                                            k += 16;
                                            continue;
                                        }
                                        k += r;
                                        if (k > Se) return fail("jpeg: AC run past the band");
                                        coef[kZigzag[k]] = extend(br.receive(sz), sz) * (1 << Al);
                                        k++;
                                    }
                                } else { // AC scan, refinement (G.1.2.3): new coefficients of magnitude 1 << Al, correction bits for the known ones
                                    const int p1 = 1 << Al, m1 = -(1 << Al);
                                    int k = Ss;
                                    auto refine = [&](int32_t& v) {
                                        if (br.bit() && (v & p1) == 0) v += v >= 0 ? p1 : m1;
                                    };
                                    if (eobrun == 0) {
                                        for (; k <= Se; k++) {
                                            int rs = 0;
                                            if (!decode_symbol(br, ac[c->ta], rs)) return fail("jpeg: bad AC code");
                                            int r = rs >> 4;
                                            const int sz = rs & 15;
                                            int value = 0;
                                            if (sz) {
                                                if (sz != 1) return fail("jpeg: bad refinement code");
                                                value = br.bit() ? p1 : m1;
                                            } else if (r != 15) {
                                                eobrun = 1 << r;
                                                if (r) eobrun += br.receive(r);
                                                break; // the rest of this block belongs to the end-of-band run
                                            }
                                            for (; k <= Se; k++) { // skip r still-zero coefficients, refining the non-zero ones on the way
                                                int32_t& v = coef[kZigzag[k]];
                                                if (v != 0) refine(v);
                                                else if (--r < 0) break;
                                            }
                                            if (sz && k <= Se) coef[kZigzag[k]] = value;
                                        }
                                    }
                                    if (eobrun > 0) {
                                        for (; k <= Se; k++) {
                                            int32_t& v = coef[kZigzag[k]];
                                            if (v != 0) refine(v);
                                        }
                                        eobrun--;
                                    }
                                }
                            }
                    }
                    if (br.hit_marker && !(my == uy - 1 && mx == ux - 1) && !(restart_interval && until_restart == 1) && !(progressive && eobrun > 0))
                        return fail("jpeg: entropy-coded data ends early");
                    if (restart_interval) until_restart--;
                }
            decoded_any = true;
            // continue behind the entropy-coded data: the next marker that is not a restart marker or a stuffed zero
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) q++;
            pos = (size_t)(q - data);
        }
        // every other segment (APPn, COM, ...) is skipped
    }
    if (!have_frame || !decoded_any) return fail("jpeg: no image data");
    for (Component& c : comps) { // dequantise, inverse DCT
        if (!qt_present[c.tq]) return fail("jpeg: component without a quantisation table");
        for (int gy = 0; gy < c.blocks_h; gy++)
            for (int gx = 0; gx < c.blocks_w; gx++) {
                int32_t* coef = c.coef.data() + ((size_t)gy * (size_t)c.blocks_w + (size_t)gx) * 64;
                for (int i = 0; i < 64; i++) coef[i] *= (int32_t)qt[c.tq][i];
                idct.run(coef, c.samples.data() + ((size_t)gy * 8 * (size_t)c.blocks_w + (size_t)gx) * 8, c.blocks_w * 8);
            }
        std::vector<int32_t>().swap(c.coef);
    }

    // ---- up-sample every component to the image grid
    const size_t npx = (size_t)W * (size_t)H;
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const Component& c = comps[ci];
        const int stride = c.blocks_w * 8;
        const int fx = hmax / c.h, fy = vmax / c.v;
        std::vector<uint8_t>& out = full[ci];
        out.resize(npx);
        auto at = [&](int x, int y) -> int {
            x = x < 0 ? 0 : (x >= c.width ? c.width - 1 : x);
            y = y < 0 ? 0 : (y >= c.height ? c.height - 1 : y);
            return c.samples[(size_t)y * (size_t)stride + (size_t)x];
        };
        const bool exact = hmax % c.h == 0 && vmax % c.v == 0;
        if (exact && fx == 1 && fy == 1) {
            for (int y = 0; y < H; y++) std::memcpy(out.data() + (size_t)y * W, c.samples.data() + (size_t)y * stride, (size_t)W);
        } else if (exact && fx == 2 && fy == 1) { // h2v1, triangle filter: 3/4 nearer + 1/4 further sample
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const int i = x >> 1;
                    int v;
                    if (x & 1) v = i + 1 < c.width ? (3 * at(i, y) + at(i + 1, y) + 2) >> 2 : at(i, y);
                    else v = i > 0 ? (3 * at(i, y) + at(i - 1, y) + 1) >> 2 : at(i, y);
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
        } else if (exact && fx == 2 && fy == 2) { // h2v2: the same filter in both directions, 16ths
            for (int y = 0; y < H; y++) {
                const int r = y >> 1, r2 = (y & 1) ? r + 1 : r - 1; // nearer and further input row
                for (int x = 0; x < W; x++) {
                    const int i = x >> 1;
                    const int cur = 3 * at(i, r) + at(i, r2);
                    int v;
                    if (x & 1) v = i + 1 < c.width ? (3 * cur + (3 * at(i + 1, r) + at(i + 1, r2)) + 7) >> 4 : (4 * cur + 7) >> 4;
                    else v = i > 0 ? (3 * cur + (3 * at(i - 1, r) + at(i - 1, r2)) + 8) >> 4 : (4 * cur + 8) >> 4;
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
            }
        } else { // any other ratio: the sample whose cell holds the pixel
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) out[(size_t)y * W + x] = (uint8_t)at(x * c.h / hmax, y * c.v / vmax);
        }
    }

    // ---- colour
    rgba.resize(npx * 4);
    if (comps.size() == 1) {
        for (size_t i = 0; i < npx; i++) { rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = full[0][i]; rgba[4 * i + 3] = 255; }
    } else {
        // three components are YCbCr unless an Adobe segment says "no transform" (or, without JFIF/Adobe hints, the ids spell "RGB")
        bool ycc = adobe_transform != 0;
        if (adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
        for (size_t i = 0; i < npx; i++) {
            int r = full[0][i], g = full[1][i], b = full[2][i];
            if (ycc) { // 16-bit fixed point, as in every libjpeg descendant: FIX(x) = x * 65536 + 0.5
                const int y = r, cb = g - 128, cr = b - 128;
                r = y + ((91881 * cr + 32768) >> 16);
                g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                b = y + ((116130 * cb + 32768) >> 16);
                r = r < 0 ? 0 : (r > 255 ? 255 : r);
                g = g < 0 ? 0 : (g > 255 ? 255 : g);
                b = b < 0 ? 0 : (b > 255 ? 255 : b);
            }
            rgba[4 * i] = (uint8_t)r; rgba[4 * i + 1] = (uint8_t)g; rgba[4 * i + 2] = (uint8_t)b; rgba[4 * i + 3] = 255;
        }
    }
    width = (uint32_t)W;
    height = (uint32_t)H;
    return true;
}

} // namespace rfw
