// gltf.cpp — glTF 2.0 (.gltf + external/embedded buffers, .glb) -> rfw::Scene.
//
// The reference imports glTF through the third-party crate l3d 0.3 (crates/rfw-scene/src/loaders/gltf.rs:26-90 calls
// l3d::LoadInstance and receives MeshDescriptors, Materials and a node list); l3d is not vendored in /root/reference, so
// this file restates what that hand-over contains, from the glTF 2.0 specification and from how the reference consumes it:
//   * every glTF mesh -> one MeshDescriptor, all TRIANGLES primitives concatenated and de-indexed (3 vertices per triangle),
//     per-vertex material id (gltf.rs:77-83 remaps them into the scene's material list), then Mesh3D::from
//     (crates/rfw-scene/src/objects_3d/mod.rs:673-895: normals generated when absent, RTTriangle lod/area, ranges);
//   * KHR_lights_punctual -> directional / point / spot lights (direction = the node's -Z axis);
//   * pbrMetallicRoughness -> Material {color, metallic, roughness}; emissiveFactor (x KHR_materials_emissive_strength)
//     replaces the colour when it is non-zero, which is how the scene recognises lights (material/list.rs:492-515: rgb > 1);
//   * the node hierarchy is flattened: every node with a mesh becomes one instance with the node's world matrix
//     (loaders/gltf.rs:86-88 + graph/mod.rs keep the hierarchy for animation; animation stays outside the backend path);
//   * JOINTS_0/WEIGHTS_0 -> JointData per vertex, skins -> SkinData with joint_matrices = world(joint) * inverseBind
//     (graph/mod.rs:592-608), and skinned instances carry the skin id; as glTF prescribes, the transform of a skinned
//     mesh's own node is ignored (instance matrix = identity).
//   * images: PNG (8-bit, non-interlaced; zlib inflates, the rest is decoded here) -> BGRA8 textures with a 5-level mip chain;
//     baseColor / normal / emissive / metallicRoughness texture references -> the material's texture ids.
// Not read: JPEG and other image formats (their texture ids stay -1), animations, sparse accessors, morph targets, cameras other
// than the first perspective one, non-triangle primitive modes.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include <zlib.h>

#include "rfw_host.hpp"

namespace rfw {
namespace {

// ---------------------------------------------------------------- JSON
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* get(const char* key) const
    {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char* key) const { return get(key) != nullptr; }
    double number(const char* key, double dflt) const
    {
        const Json* j = get(key);
        return j && j->kind == Num ? j->num : dflt;
    }
    int64_t integer(const char* key, int64_t dflt) const
    {
        const Json* j = get(key);
        return j && j->kind == Num ? (int64_t)j->num : dflt;
    }
    std::string string(const char* key, const std::string& dflt = "") const
    {
        const Json* j = get(key);
        return j && j->kind == Str ? j->str : dflt;
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
};

struct JsonParser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;

    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool fail(const char* m)
    {
        if (err.empty()) err = m;
        return false;
    }
    bool parse_string(std::string& out)
    {
        if (p >= end || *p != '"') return fail("json: expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("json: bad escape");
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (end - p < 5) return fail("json: bad \\u escape");
                    unsigned v = 0;
                    for (int k = 1; k <= 4; k++) {
                        const char c = p[k];
                        v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0)));
                    }
                    p += 4;
                    if (v < 0x80) out += (char)v;
                    else if (v < 0x800) { out += (char)(0xC0 | (v >> 6)); out += (char)(0x80 | (v & 0x3F)); }
                    else { out += (char)(0xE0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 0x3F)); out += (char)(0x80 | (v & 0x3F)); }
                    break;
                }
                default: out += *p; break;
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("json: unterminated string");
        p++;
        return true;
    }
    bool parse(Json& out)
    {
        if (++depth > 64) return fail("json: nesting too deep");
        ws();
        if (p >= end) return fail("json: unexpected end");
        bool ok = true;
        if (*p == '{') {
            out.kind = Json::Obj;
            p++;
            ws();
            if (p < end && *p == '}') { p++; }
            else {
                for (;;) {
                    ws();
                    std::string key;
                    if (!parse_string(key)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("json: expected ':'"); break; }
                    p++;
                    out.obj.emplace_back(key, Json());
                    if (!parse(out.obj.back().second)) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == '}') { p++; break; }
                    ok = fail("json: expected ',' or '}'");
                    break;
                }
            }
        } else if (*p == '[') {
            out.kind = Json::Arr;
            p++;
            ws();
            if (p < end && *p == ']') { p++; }
            else {
                for (;;) {
                    out.arr.emplace_back();
                    if (!parse(out.arr.back())) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == ']') { p++; break; }
                    ok = fail("json: expected ',' or ']'");
                    break;
                }
            }
        } else if (*p == '"') {
            out.kind = Json::Str;
            ok = parse_string(out.str);
        } else if (end - p >= 4 && !std::strncmp(p, "true", 4)) { out.kind = Json::Bool; out.b = true; p += 4; }
        else if (end - p >= 5 && !std::strncmp(p, "false", 5)) { out.kind = Json::Bool; out.b = false; p += 5; }
        else if (end - p >= 4 && !std::strncmp(p, "null", 4)) { out.kind = Json::Null; p += 4; }
        else {
            const char* q = p;
            while (q < end && *q != '\0' && (std::strchr("+-0123456789.eE", *q) != nullptr)) q++;
            if (q == p) ok = fail("json: unexpected character");
            else {
                const std::string tok(p, q);
                char* e = nullptr;
                out.kind = Json::Num;
                out.num = std::strtod(tok.c_str(), &e);
                if (e == tok.c_str()) ok = fail("json: bad number");
                p = q;
            }
        }
        depth--;
        return ok;
    }
};

// ---------------------------------------------------------------- small helpers
bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read(reinterpret_cast<char*>(out.data()), n);
    return (bool)f;
}

bool base64_decode(const std::string& s, size_t from, std::vector<uint8_t>& out)
{
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); i++) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=') break;
        else if (c == '\n' || c == '\r' || c == ' ') continue;
        else return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)((acc >> bits) & 0xffu));
        }
    }
    return true;
}

std::string dir_of(const std::string& path)
{
    const size_t k = path.find_last_of("/\\");
    return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}

// column-major 4x4 helpers (double for the hierarchy, rounded once at the end)
struct M4 {
    double m[16];
};
M4 m4_identity()
{
    M4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}
M4 m4_mul(const M4& a, const M4& b)
{
    M4 r{};
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
M4 m4_trs(const double t[3], const double q[4], const double s[3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r = m4_identity();
    r.m[0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[1] = (2 * (x * y + z * w)) * s[0]; r.m[2] = (2 * (x * z - y * w)) * s[0];
    r.m[4] = (2 * (x * y - z * w)) * s[1]; r.m[5] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[6] = (2 * (y * z + x * w)) * s[1];
    r.m[8] = (2 * (x * z + y * w)) * s[2]; r.m[9] = (2 * (y * z - x * w)) * s[2]; r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
    return r;
}
rfw_mat4 to_f32(const M4& a)
{
    rfw_mat4 r;
    for (int i = 0; i < 16; i++) r.m[i] = (float)a.m[i];
    return r;
}

// ---------------------------------------------------------------- PNG (8-bit, non-interlaced) -> RGBA8
// The inflate step is zlib's; chunk walk, scanline filters (PNG 1.2 §6) and the expansion to RGBA are below.
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

bool decode_png(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) { err = "png: bad signature"; return false; }
    size_t off = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool end = false;
    while (!end && off + 12 <= size) {
        const uint32_t len = be32(data + off);
        const uint8_t* type = data + off + 4;
        const uint8_t* body = data + off + 8;
        if (off + 12 + (size_t)len > size) { err = "png: chunk runs past the end"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "png: short IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; colour = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        off += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || (uint64_t)w * h > (1ull << 26)) { err = "png: bad dimensions"; return false; }
    int channels;
    switch (colour) {
    case 0: channels = 1; break; // grey
    case 2: channels = 3; break; // rgb
    case 3: channels = 1; break; // palette
    case 4: channels = 2; break; // grey + alpha
    case 6: channels = 4; break; // rgba
    default: err = "png: unknown colour type"; return false;
    }
    const bool depth_ok = colour == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : colour == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok || interlace > 1) { err = "png: bad bit depth or interlace method"; return false; }
    const size_t bits_pp = (size_t)channels * (size_t)depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1; // filter distance in bytes
    // the image is one pass, or the seven passes of Adam7 (PNG specification, section 8.2): start and step of each pass in x and y
    static const int px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Pass { uint32_t w, h; int x0, y0, dx, dy; size_t stride; };
    std::vector<Pass> passes;
    size_t total = 0;
    for (int k = 0; k < (interlace ? 7 : 1); k++) {
        Pass q;
        q.x0 = interlace ? px0[k] : 0; q.y0 = interlace ? py0[k] : 0; q.dx = interlace ? pdx[k] : 1; q.dy = interlace ? pdy[k] : 1;
        q.w = (w + (uint32_t)q.dx - 1 - (uint32_t)q.x0) / (uint32_t)q.dx;
        q.h = (h + (uint32_t)q.dy - 1 - (uint32_t)q.y0) / (uint32_t)q.dy;
        if ((uint32_t)q.x0 >= w || (uint32_t)q.y0 >= h) q.w = q.h = 0;
        q.stride = ((size_t)q.w * bits_pp + 7) / 8;
        if (q.w && q.h) total += (q.stride + 1) * q.h;
        passes.push_back(q);
    }
    std::vector<uint8_t> raw(total);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { err = "png: inflate failed"; return false; }
    // colour-key transparency of grey / rgb images (tRNS holds one 16-bit sample per channel)
    int key[3] = {-1, -1, -1};
    if (colour == 0 && trns.size() >= 2) key[0] = (trns[0] << 8) | trns[1];
    if (colour == 2 && trns.size() >= 6) for (int c = 0; c < 3; c++) key[c] = (trns[2 * c] << 8) | trns[2 * c + 1];
    rgba.assign((size_t)w * h * 4, 0);
    size_t off_raw = 0;
    std::vector<uint8_t> prev, cur;
    for (const Pass& q : passes) {
        if (!q.w || !q.h) continue;
        prev.assign(q.stride, 0);
        cur.resize(q.stride);
        for (uint32_t y = 0; y < q.h; y++) {
            const uint8_t filter = raw[off_raw];
            const uint8_t* src = raw.data() + off_raw + 1;
            off_raw += q.stride + 1;
            for (size_t x = 0; x < q.stride; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int v = src[x];
                switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = "png: bad filter"; return false;
                }
                cur[x] = (uint8_t)v;
            }
            // sample s of pixel x as the file stores it (depth bits, most significant bits first)
            auto sample = [&](uint32_t x, int s) -> int {
                const size_t idx = (size_t)x * (size_t)channels + (size_t)s;
                if (depth == 8) return cur[idx];
                if (depth == 16) return (cur[2 * idx] << 8) | cur[2 * idx + 1];
                const size_t bit = idx * (size_t)depth;
                return (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
            };
            auto to8 = [&](int v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (depth == 8 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1))); };
            for (uint32_t x = 0; x < q.w; x++) {
                uint8_t r, g, bl, al = 255;
                switch (colour) {
                case 0: { const int v = sample(x, 0); r = g = bl = to8(v); if (v == key[0]) al = 0; break; }
                case 2: {
                    const int v0 = sample(x, 0), v1 = sample(x, 1), v2 = sample(x, 2);
                    r = to8(v0); g = to8(v1); bl = to8(v2);
                    if (v0 == key[0] && v1 == key[1] && v2 == key[2]) al = 0;
                    break;
                }
                case 3: {
                    const size_t k = (size_t)sample(x, 0);
                    if (3 * k + 2 >= palette.size()) { err = "png: palette index out of range"; return false; }
                    r = palette[3 * k]; g = palette[3 * k + 1]; bl = palette[3 * k + 2];
                    if (k < trns.size()) al = trns[k];
                    break;
                }
                case 4: r = g = bl = to8(sample(x, 0)); al = to8(sample(x, 1)); break;
                default: r = to8(sample(x, 0)); g = to8(sample(x, 1)); bl = to8(sample(x, 2)); al = to8(sample(x, 3)); break;
                }
                uint8_t* o = rgba.data() + ((size_t)((uint32_t)q.y0 + y * (uint32_t)q.dy) * w + (size_t)((uint32_t)q.x0 + x * (uint32_t)q.dx)) * 4;
                o[0] = r; o[1] = g; o[2] = bl; o[3] = al;
            }
            prev.swap(cur);
            cur.resize(q.stride);
        }
    }
    width = w; height = h;
    return true;
}

// ---------------------------------------------------------------- the document
struct Doc {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string err;
    std::string dir;

    bool fail(const std::string& m)
    {
        if (err.empty()) err = m;
        return false;
    }

    static int components(const std::string& type)
    {
        if (type == "SCALAR") return 1;
        if (type == "VEC2") return 2;
        if (type == "VEC3") return 3;
        if (type == "VEC4") return 4;
        if (type == "MAT4") return 16;
        return 0;
    }
    static int component_bytes(int64_t ct)
    {
        switch (ct) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
        default: return 0;
        }
    }

    // accessor -> count x ncomp doubles (integers converted; normalised integers mapped to [0,1] / [-1,1] when `normalise`)
    bool read_accessor(int64_t index, int want_comp, bool normalise, std::vector<double>& out, size_t& count)
    {
        const Json* accs = root.get("accessors");
        if (!accs || index < 0 || (size_t)index >= accs->size()) return fail("gltf: accessor index out of range");
        const Json& a = accs->arr[(size_t)index];
        if (a.has("sparse")) return fail("gltf: sparse accessors are not supported");
        const int nc = components(a.string("type"));
        const int64_t ct = a.integer("componentType", 0);
        const int cb = component_bytes(ct);
        if (nc == 0 || cb == 0) return fail("gltf: unsupported accessor type");
        if (want_comp && nc != want_comp) return fail("gltf: accessor has the wrong number of components");
        const int64_t declared = a.integer("count", 0);
        if (declared < 0 || declared > (int64_t)1 << 31) return fail("gltf: accessor count out of range");
        count = (size_t)declared;
        const bool norm = normalise || (a.get("normalized") && a.get("normalized")->b);
        const int64_t bv_index = a.integer("bufferView", -1);
        if (bv_index < 0) { // all zeros
            if (count > ((size_t)1 << 26)) return fail("gltf: accessor without a buffer view is too large");
            out.assign(count * (size_t)nc, 0.0);
            return true;
        }
        const Json* bvs = root.get("bufferViews");
        if (!bvs || (size_t)bv_index >= bvs->size()) return fail("gltf: bufferView index out of range");
        const Json& bv = bvs->arr[(size_t)bv_index];
        const int64_t bi = bv.integer("buffer", -1);
        if (bi < 0 || (size_t)bi >= buffers.size()) return fail("gltf: buffer index out of range");
        const std::vector<uint8_t>& buf = buffers[(size_t)bi];
        const size_t elem = (size_t)nc * (size_t)cb;
        // every field is checked by itself first: a negative or enormous offset must not wrap the sums below into range
        const int64_t view_off = bv.integer("byteOffset", 0), acc_off = a.integer("byteOffset", 0), view_bytes = bv.integer("byteLength", 0), view_stride = bv.integer("byteStride", 0);
        if (view_off < 0 || acc_off < 0 || view_bytes < 0 || view_stride < 0 || view_stride > 65536 || (uint64_t)view_off > buf.size() || (uint64_t)view_bytes > buf.size() - (size_t)view_off ||
            acc_off > view_bytes)
            return fail("gltf: buffer view or accessor offset out of range");
        size_t stride = (size_t)view_stride;
        if (stride == 0) stride = elem;
        const size_t base = (size_t)view_off + (size_t)acc_off;
        const size_t view_len = (size_t)view_bytes, in_view = (size_t)acc_off;
        if (count && (stride < elem || in_view + (count - 1) * stride + elem > view_len || base + (count - 1) * stride + elem > buf.size()))
            return fail("gltf: accessor reads past the end of its buffer view");
        out.assign(count * (size_t)nc, 0.0); // only now: the count is backed by bytes that exist
        for (size_t i = 0; i < count; i++) {
            const uint8_t* src = buf.data() + base + i * stride;
            for (int c = 0; c < nc; c++) {
                double v = 0.0;
                switch (ct) {
                case 5120: { int8_t x; std::memcpy(&x, src + c, 1); v = norm ? std::max((double)x / 127.0, -1.0) : (double)x; break; }
                case 5121: { uint8_t x; std::memcpy(&x, src + c, 1); v = norm ? (double)x / 255.0 : (double)x; break; }
                case 5122: { int16_t x; std::memcpy(&x, src + 2 * c, 2); v = norm ? std::max((double)x / 32767.0, -1.0) : (double)x; break; }
                case 5123: { uint16_t x; std::memcpy(&x, src + 2 * c, 2); v = norm ? (double)x / 65535.0 : (double)x; break; }
                case 5125: { uint32_t x; std::memcpy(&x, src + 4 * c, 4); v = (double)x; break; }
                default: { float x; std::memcpy(&x, src + 4 * c, 4); v = (double)x; break; }
                }
                out[i * (size_t)nc + (size_t)c] = v;
            }
        }
        return true;
    }
};

bool open_document(const std::string& path, Doc& doc)
{
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return doc.fail("gltf: cannot read " + path);
    std::vector<uint8_t> glb_bin;
    bool have_glb_bin = false;
    const char* json_begin = reinterpret_cast<const char*>(file.data());
    size_t json_len = file.size();
    if (file.size() >= 12 && !std::memcmp(file.data(), "glTF", 4)) { // binary container: header, JSON chunk, optional BIN chunk
        uint32_t version, total;
        std::memcpy(&version, file.data() + 4, 4);
        std::memcpy(&total, file.data() + 8, 4);
        if (version != 2 || total > file.size()) return doc.fail("gltf: bad GLB header");
        size_t off = 12;
        bool have_json = false;
        while (off + 8 <= total) {
            uint32_t len, type;
            std::memcpy(&len, file.data() + off, 4);
            std::memcpy(&type, file.data() + off + 4, 4);
            off += 8;
            if (off + len > total) return doc.fail("gltf: GLB chunk runs past the end of the file");
            if (type == 0x4E4F534Au && !have_json) { json_begin = reinterpret_cast<const char*>(file.data() + off); json_len = len; have_json = true; }
            else if (type == 0x004E4942u && !have_glb_bin) { glb_bin.assign(file.begin() + (long)off, file.begin() + (long)(off + len)); have_glb_bin = true; }
            off += (len + 3u) & ~3u;
        }
        if (!have_json) return doc.fail("gltf: GLB without a JSON chunk");
    }
    JsonParser jp{json_begin, json_begin + json_len, std::string()};
    if (!jp.parse(doc.root) || doc.root.kind != Json::Obj) return doc.fail(jp.err.empty() ? "gltf: the document is not a JSON object" : jp.err);
    const Json* asset = doc.root.get("asset");
    if (!asset || asset->string("version").substr(0, 1) != "2") return doc.fail("gltf: asset.version 2.x required");
    const Json* bufs = doc.root.get("buffers");
    const std::string dir = dir_of(path);
    doc.dir = dir;
    for (size_t i = 0; bufs && i < bufs->size(); i++) {
        const Json& b = bufs->arr[i];
        std::vector<uint8_t> data;
        const std::string uri = b.string("uri");
        if (uri.empty()) {
            if (i != 0 || !have_glb_bin) return doc.fail("gltf: buffer without uri outside a GLB");
            data = glb_bin;
        } else if (uri.compare(0, 5, "data:") == 0) {
            const size_t comma = uri.find(',');
            if (comma == std::string::npos || !base64_decode(uri, comma + 1, data)) return doc.fail("gltf: bad data: uri");
        } else {
            if (uri.find("..") != std::string::npos || uri[0] == '/') return doc.fail("gltf: buffer uri must stay below the document's directory");
            if (!read_file(dir + uri, data)) return doc.fail("gltf: cannot read buffer " + uri);
        }
        if (data.size() < (size_t)b.integer("byteLength", 0)) return doc.fail("gltf: buffer shorter than its byteLength");
        doc.buffers.push_back(std::move(data));
    }
    return true;
}

M4 m4_from(const double* v) { M4 r; std::memcpy(r.m, v, sizeof(r.m)); return r; }
M4 m4_from_f32(const rfw_mat4& a) { M4 r; for (int i = 0; i < 16; i++) r.m[i] = (double)a.m[i]; return r; }
M4 local_matrix(const GraphNode& n) { return n.has_matrix ? m4_from(n.matrix) : m4_trs(n.translation, n.rotation, n.scale); }

// value of one sampler at time t (glTF 2.0 section 3.11 / appendix C): nc components, rotations (nc == 4) by spherical interpolation
void sample_channel(const AnimationSampler& s, int nc, double t, double* out)
{
    const size_t keys = s.times.size();
    const size_t per_key = (size_t)nc * (s.interpolation == 2 ? 3u : 1u), value_at = s.interpolation == 2 ? (size_t)nc : 0u;
    auto value = [&](size_t k) { return s.values.data() + k * per_key + value_at; };
    auto finish = [&]() {
        if (nc != 4) return;
        const double len = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
        if (len > 0.0) for (int c = 0; c < 4; c++) out[c] /= len;
    };
    if (t <= s.times.front() || keys == 1) { for (int c = 0; c < nc; c++) out[c] = value(0)[c]; finish(); return; }
    if (t >= s.times.back()) { for (int c = 0; c < nc; c++) out[c] = value(keys - 1)[c]; finish(); return; }
    size_t k = (size_t)(std::upper_bound(s.times.begin(), s.times.end(), t) - s.times.begin()) - 1; // times[k] <= t < times[k + 1]
    const double dt = s.times[k + 1] - s.times[k], u = dt > 0.0 ? (t - s.times[k]) / dt : 0.0;
    const double* a = value(k);
    const double* b = value(k + 1);
    if (s.interpolation == 1) { for (int c = 0; c < nc; c++) out[c] = a[c]; finish(); return; }
    if (s.interpolation == 2) { // cubic Hermite spline: out-tangent of key k, in-tangent of key k + 1, both scaled by the key distance
        const double* bk = s.values.data() + k * per_key + 2 * (size_t)nc;
        const double* ak1 = s.values.data() + (k + 1) * per_key;
        const double u2 = u * u, u3 = u2 * u;
        for (int c = 0; c < nc; c++)
            out[c] = (2.0 * u3 - 3.0 * u2 + 1.0) * a[c] + dt * (u3 - 2.0 * u2 + u) * bk[c] + (-2.0 * u3 + 3.0 * u2) * b[c] + dt * (u3 - u2) * ak1[c];
        finish();
        return;
    }
    if (nc == 4) { // slerp along the shorter arc; nearly parallel quaternions interpolate linearly
        double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        const double sign = d < 0.0 ? -1.0 : 1.0;
        d = std::fabs(d);
        double wa = 1.0 - u, wb = u;
        if (d <= 0.9995) {
            const double theta = std::acos(d), st = std::sin(theta);
            wa = std::sin((1.0 - u) * theta) / st;
            wb = std::sin(u * theta) / st;
        }
        for (int c = 0; c < 4; c++) out[c] = wa * a[c] + wb * sign * b[c];
        finish();
        return;
    }
    for (int c = 0; c < nc; c++) out[c] = a[c] + u * (b[c] - a[c]);
}

} // namespace

bool decode_image(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err)
{
    if (size >= 8 && data[0] == 0x89 && data[1] == 'P' && data[2] == 'N' && data[3] == 'G') return decode_png(data, size, width, height, rgba, err);
    if (size >= 3 && data[0] == 0xFF && data[1] == 0xD8) return decode_jpeg(data, size, width, height, rgba, err);
    err = "image: neither PNG nor JPEG";
    return false;
}

// graph/mod.rs:540-630 traverse_children, restated on the flattened parents-first order: combined = parent's combined x local.  A skinned
// instance keeps the identity as its instance matrix and its joints carry the whole transform, joint = combined(joint node) x inverse bind
// (glTF 2.0 section 3.7.3.3: the skinned mesh node's own transform is ignored) — the reference hands the node's combined matrix to the
// instance and multiplies the joints by its inverse (:587-600): the same product up to rounding, without the 4x4 inverse.
void NodeGraph::update(Scene& scene)
{
    bool lights_stale = false;
    for (const uint32_t ni : order) {
        GraphNode& n = nodes[ni];
        const M4 parent = parent_of[ni] == UINT32_MAX ? m4_from(root) : m4_from(nodes[parent_of[ni]].world);
        const M4 w = m4_mul(parent, local_matrix(n));
        std::memcpy(n.world, w.m, sizeof(n.world));
    }
    for (const uint32_t ni : order) {
        const GraphNode& n = nodes[ni];
        if (n.mesh < 0 || n.instance < 0 || n.skin >= 0) continue;
        auto it = scene.instances_3d.find((uint32_t)n.mesh);
        if (it == scene.instances_3d.end() || (size_t)n.instance >= it->second.matrices.size()) continue;
        const rfw_mat4 m = to_f32(m4_from(n.world));
        if (!std::memcmp(&m, &it->second.matrices[(size_t)n.instance], sizeof(m))) continue;
        scene.set_matrix((uint32_t)n.mesh, (size_t)n.instance, m);
        auto mit = scene.meshes_3d.find((uint32_t)n.mesh);
        if (mit != scene.meshes_3d.end())
            for (const rfw_vertex_mesh& r : mit->second.ranges)
                lights_stale = lights_stale || (r.mat_id < scene.materials.size() && is_emissive(scene.materials[r.mat_id]));
    }
    for (const auto& sk : skins) {
        if (sk.first < 0 || (size_t)sk.first >= scene.skins.size()) continue;
        Skin& skin = scene.skins[(size_t)sk.first];
        for (size_t j = 0; j < sk.second.size() && j < skin.joint_matrices.size(); j++) {
            const int64_t jn = sk.second[j];
            const M4 jw = (jn >= 0 && (size_t)jn < nodes.size()) ? m4_from(nodes[(size_t)jn].world) : m4_identity();
            skin.joint_matrices[j] = to_f32(m4_mul(jw, m4_from_f32(skin.inverse_bind_matrices[j])));
        }
        scene.skins_changed = true;
    }
    if (lights_stale) scene.update_lights(); // an emitter moved: its area lights are world-space triangles (lib.rs:575-648)
}

void NodeGraph::set_root_transform(Scene& scene, const double translation[3], const double rotation_xyzw[4], const double scale[3])
{
    const M4 r = m4_trs(translation, rotation_xyzw, scale);
    std::memcpy(root, r.m, sizeof(root));
    update(scene);
}

void NodeGraph::set_animation_time(Scene& scene, double time)
{
    if (active_animation < 0 || (size_t)active_animation >= animations.size()) return;
    const Animation& a = animations[(size_t)active_animation];
    double t = 0.0;
    if (a.duration > 0.0) { // the animation loops
        t = std::fmod(time, a.duration);
        if (t < 0.0) t += a.duration;
    }
    for (const AnimationChannel& ch : a.channels) {
        if (ch.node >= nodes.size() || ch.sampler >= a.samplers.size()) continue;
        GraphNode& n = nodes[ch.node];
        const AnimationSampler& s = a.samplers[ch.sampler];
        if (n.has_matrix || s.times.empty()) continue;
        double* dst = ch.path == 0 ? n.translation : (ch.path == 1 ? n.rotation : n.scale);
        sample_channel(s, ch.path == 1 ? 4 : 3, t, dst);
    }
    update(scene);
}

size_t Scene::instantiate_graph(size_t graph)
{
    if (graph >= graphs.size()) return graphs.size();
    NodeGraph g = graphs[graph]; // nodes, animations, clock and transform copied
    std::map<int32_t, int32_t> new_skin; // skin of the original -> its copy
    for (auto& sk : g.skins) {
        if (sk.first < 0 || (size_t)sk.first >= skins.size()) continue;
        const Skin copy = skins[(size_t)sk.first];
        new_skin[sk.first] = (int32_t)skins.size();
        sk.first = (int32_t)skins.size();
        skins.push_back(copy);
    }
    for (GraphNode& n : g.nodes) {
        if (n.mesh < 0 || n.instance < 0) continue;
        const uint32_t mesh = (uint32_t)n.mesh;
        const bool skinned = n.skin >= 0 && new_skin.count(n.skin);
        const size_t slot = add_instance(mesh, skinned ? mat4_identity() : to_f32(m4_from(n.world)));
        n.instance = (int64_t)slot;
        if (skinned) {
            InstanceList3D& l = instances_3d[mesh];
            if (l.skin_ids.size() <= slot) l.skin_ids.resize(slot + 1, -1);
            n.skin = new_skin[n.skin];
            l.skin_ids[slot] = n.skin;
        }
    }
    skins_changed = skins_changed || !new_skin.empty();
    graphs.push_back(std::move(g));
    graphs.back().update(*this);
    update_lights(); // a copied emitter is a new set of area lights (as at the end of load_gltf)
    return graphs.size() - 1;
}

void Scene::set_animations_time(double time)
{
    for (NodeGraph& g : graphs) g.set_animation_time(*this, time);
}

// Adds the document's materials, meshes, instances and skins to `scene` (on top of what it already holds) and, when the
// document has a perspective camera and `cam` is given, aims `cam` like it.  Returns false and sets `err` on any malformed input.
bool load_gltf(const std::string& path, Scene& scene, Camera3D* cam, std::string& err)
{
    Doc doc;
    if (!open_document(path, doc)) { err = doc.err; return false; }
    const Json& root = doc.root;
    auto arr = [&](const char* key) -> const std::vector<Json>& {
        static const std::vector<Json> empty;
        const Json* j = root.get(key);
        return j && j->kind == Json::Arr ? j->arr : empty;
    };

    // ---- images -> scene textures (PNG and JPEG: BGRA8 with a 5-level mip chain, what l3d hands the trait); texture -> scene texture id
    std::vector<int> image_tex; // glTF image index -> scene texture id, -1 = not readable here (KTX, WebP, ...)
    for (const Json& im : arr("images")) {
        std::vector<uint8_t> file;
        const std::string uri = im.string("uri");
        bool have = false;
        if (!uri.empty()) {
            if (uri.compare(0, 5, "data:") == 0) {
                const size_t comma = uri.find(',');
                have = comma != std::string::npos && base64_decode(uri, comma + 1, file);
            } else if (uri.find("..") == std::string::npos && uri[0] != '/') {
                have = read_file(doc.dir + uri, file);
            }
        } else if (im.has("bufferView")) {
            const Json* bvs = root.get("bufferViews");
            const int64_t bi = im.integer("bufferView", -1);
            if (bvs && bi >= 0 && (size_t)bi < bvs->size()) {
                const Json& bv = bvs->arr[(size_t)bi];
                const int64_t b = bv.integer("buffer", -1);
                const int64_t o64 = bv.integer("byteOffset", 0), n64 = bv.integer("byteLength", 0);
                const size_t o = (size_t)o64, n = (size_t)n64;
                if (b >= 0 && (size_t)b < doc.buffers.size() && o64 >= 0 && n64 >= 0 && o <= doc.buffers[(size_t)b].size() && n <= doc.buffers[(size_t)b].size() - o) {
                    file.assign(doc.buffers[(size_t)b].begin() + (long)o, doc.buffers[(size_t)b].begin() + (long)(o + n));
                    have = true;
                }
            }
        }
        int tex_id = -1;
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> rgba;
        std::string perr;
        if (have && decode_image(file.data(), file.size(), w, h, rgba, perr)) {
            Texture t;
            t.width = w; t.height = h; t.format = RFW_FORMAT_BGRA8;
            t.bytes.resize(rgba.size());
            for (size_t i = 0; i < (size_t)w * h; i++) { t.bytes[4 * i] = rgba[4 * i + 2]; t.bytes[4 * i + 1] = rgba[4 * i + 1]; t.bytes[4 * i + 2] = rgba[4 * i]; t.bytes[4 * i + 3] = rgba[4 * i + 3]; }
            t.generate_mipmaps(5);
            tex_id = (int)scene.textures.size();
            scene.textures.push_back(std::move(t));
            scene.textures_changed = true;
        }
        image_tex.push_back(tex_id);
    }
    auto texture_of = [&](const Json* ref) -> int { // {"index": glTF texture} -> scene texture id
        if (!ref) return -1;
        const int64_t ti = ref->integer("index", -1);
        const std::vector<Json>& texs = arr("textures");
        if (ti < 0 || (size_t)ti >= texs.size()) return -1;
        const int64_t src = texs[(size_t)ti].integer("source", -1);
        return (src >= 0 && (size_t)src < image_tex.size()) ? image_tex[(size_t)src] : -1;
    };

    // ---- materials (pbrMetallicRoughness; index materials.size() = the glTF default material, created on demand)
    std::vector<uint32_t> mat_ids;
    for (const Json& m : arr("materials")) {
        Material mat;
        mat.metallic = 1.0f; mat.roughness = 1.0f; // glTF defaults
        if (const Json* pbr = m.get("pbrMetallicRoughness")) {
            if (const Json* c = pbr->get("baseColorFactor"))
                for (size_t k = 0; k < 4 && k < c->size(); k++) mat.color[k] = (float)c->arr[k].num;
            mat.metallic = (float)pbr->number("metallicFactor", 1.0);
            mat.roughness = (float)pbr->number("roughnessFactor", 1.0);
            mat.diffuse_tex = texture_of(pbr->get("baseColorTexture"));
            mat.metallic_roughness_tex = texture_of(pbr->get("metallicRoughnessTexture"));
        }
        mat.normal_tex = texture_of(m.get("normalTexture"));
        mat.emissive_tex = texture_of(m.get("emissiveTexture"));
        double strength = 1.0;
        if (const Json* ext = m.get("extensions"))
            if (const Json* es = ext->get("KHR_materials_emissive_strength")) strength = es->number("emissiveStrength", 1.0);
        if (const Json* e = m.get("emissiveFactor")) {
            double ev[3] = {0, 0, 0};
            for (size_t k = 0; k < 3 && k < e->size(); k++) ev[k] = e->arr[k].num * strength;
            if (ev[0] > 0.0 || ev[1] > 0.0 || ev[2] > 0.0) // an emitter: radiance goes into the colour (material/list.rs:492-515 keys on rgb > 1)
                for (int k = 0; k < 3; k++) mat.color[k] = (float)ev[k];
        }
        if (const Json* ext = m.get("extensions")) {
            if (const Json* ior = ext->get("KHR_materials_ior")) mat.eta = (float)ior->number("ior", 1.5);
            if (const Json* tr = ext->get("KHR_materials_transmission")) mat.transmission = (float)tr->number("transmissionFactor", 0.0);
        }
        // a file written by save_glb (gltf_export.cpp) carries the Disney parameters glTF has no slot for, and exact emitter colours
        if (const Json* ex = m.get("extras"))
            if (const Json* d = ex->get("rfw_disney")) {
                auto vec4 = [&](const char* k, float* dst) {
                    if (const Json* a = d->get(k))
                        for (size_t i = 0; i < 4 && i < a->size(); i++) dst[i] = (float)a->arr[i].num;
                };
                vec4("color", mat.color); vec4("absorption", mat.absorption); vec4("specular", mat.specular);
                if (const Json* pa = d->get("params"); pa && pa->size() >= 13) {
                    float* dst[13] = {&mat.subsurface, &mat.specular_f, &mat.specular_tint, &mat.anisotropic, &mat.sheen, &mat.sheen_tint, &mat.clearcoat,
                                      &mat.clearcoat_gloss, &mat.eta, &mat.custom0, &mat.custom1, &mat.custom2, &mat.custom3};
                    for (size_t i = 0; i < 13; i++) *dst[i] = (float)pa->arr[i].num;
                }
            }
        mat_ids.push_back(scene.add_material(mat));
    }
    int64_t default_mat = -1;
    auto material_of = [&](int64_t gltf_index) -> int32_t {
        if (gltf_index >= 0 && (size_t)gltf_index < mat_ids.size()) return (int32_t)mat_ids[(size_t)gltf_index];
        if (default_mat < 0) {
            Material mat;
            mat.metallic = 1.0f; mat.roughness = 1.0f;
            default_mat = scene.add_material(mat);
        }
        return (int32_t)default_mat;
    };

    // ---- meshes
    std::vector<int64_t> mesh_ids; // glTF mesh index -> scene mesh id (-1 = no triangles)
    for (const Json& m : arr("meshes")) {
        MeshDescriptor desc;
        desc.name = m.string("name");
        std::vector<rfw_joint_data> joints;
        bool any_normals = false, any_tangents = false, all_skinned = true;
        const Json* prims = m.get("primitives");
        for (size_t pi = 0; prims && pi < prims->size(); pi++) {
            const Json& prim = prims->arr[pi];
            if (prim.integer("mode", 4) != 4) continue; // points / lines / strips / fans are not part of the path
            const Json* attrs = prim.get("attributes");
            if (!attrs || !attrs->has("POSITION")) continue;
            std::vector<double> pos, nor, uv, tan, jnt, wgt, idx;
            size_t n_pos = 0, n = 0;
            if (!doc.read_accessor(attrs->integer("POSITION", -1), 3, false, pos, n_pos)) { err = doc.err; return false; }
            const bool has_n = attrs->has("NORMAL"), has_uv = attrs->has("TEXCOORD_0"), has_t = attrs->has("TANGENT");
            const bool has_j = attrs->has("JOINTS_0") && attrs->has("WEIGHTS_0");
            if (has_n && (!doc.read_accessor(attrs->integer("NORMAL", -1), 3, false, nor, n) || n != n_pos)) { err = doc.err.empty() ? "gltf: NORMAL count differs from POSITION" : doc.err; return false; }
            if (has_uv && (!doc.read_accessor(attrs->integer("TEXCOORD_0", -1), 2, true, uv, n) || n != n_pos)) { err = doc.err.empty() ? "gltf: TEXCOORD_0 count differs from POSITION" : doc.err; return false; }
            if (has_t && (!doc.read_accessor(attrs->integer("TANGENT", -1), 4, false, tan, n) || n != n_pos)) { err = doc.err.empty() ? "gltf: TANGENT count differs from POSITION" : doc.err; return false; }
            if (has_j) {
                if (!doc.read_accessor(attrs->integer("JOINTS_0", -1), 4, false, jnt, n) || n != n_pos) { err = doc.err.empty() ? "gltf: JOINTS_0 count differs from POSITION" : doc.err; return false; }
                if (!doc.read_accessor(attrs->integer("WEIGHTS_0", -1), 4, true, wgt, n) || n != n_pos) { err = doc.err.empty() ? "gltf: WEIGHTS_0 count differs from POSITION" : doc.err; return false; }
            }
            size_t n_idx = 0;
            if (prim.has("indices")) {
                if (!doc.read_accessor(prim.integer("indices", -1), 1, false, idx, n_idx)) { err = doc.err; return false; }
            } else {
                n_idx = n_pos;
                idx.resize(n_pos);
                for (size_t i = 0; i < n_pos; i++) idx[i] = (double)i;
            }
            n_idx -= n_idx % 3;
            const int32_t mat = material_of(prim.integer("material", -1));
            // earlier primitives of this mesh without the optional attribute get zeros (normals: the whole mesh is then regenerated)
            if (has_n && !any_normals) { desc.normals.assign(desc.vertices.size(), rfw_vec3{0, 0, 0}); any_normals = true; }
            if (has_t && !any_tangents) { desc.tangents.assign(desc.vertices.size(), rfw_vec4{0, 0, 0, 0}); any_tangents = true; }
            for (size_t k = 0; k < n_idx; k++) {
                const double di = idx[k];
                if (!(di >= 0.0) || (size_t)di >= n_pos) { err = "gltf: index out of range"; return false; }
                const size_t i = (size_t)di;
                desc.vertices.push_back(rfw_vec4{(float)pos[3 * i], (float)pos[3 * i + 1], (float)pos[3 * i + 2], 1.0f});
                if (any_normals) desc.normals.push_back(has_n ? rfw_vec3{(float)nor[3 * i], (float)nor[3 * i + 1], (float)nor[3 * i + 2]} : rfw_vec3{0, 0, 0});
                desc.uvs.push_back(has_uv ? rfw_vec2{(float)uv[2 * i], (float)uv[2 * i + 1]} : rfw_vec2{0, 0});
                if (any_tangents) desc.tangents.push_back(has_t ? rfw_vec4{(float)tan[4 * i], (float)tan[4 * i + 1], (float)tan[4 * i + 2], (float)tan[4 * i + 3]} : rfw_vec4{0, 0, 0, 0});
                desc.material_ids.push_back(mat);
                rfw_joint_data jd{};
                if (has_j) {
                    for (int c = 0; c < 4; c++) jd.joint[c] = (uint32_t)jnt[4 * i + (size_t)c];
                    jd.weight = rfw_vec4{(float)wgt[4 * i], (float)wgt[4 * i + 1], (float)wgt[4 * i + 2], (float)wgt[4 * i + 3]};
                }
                joints.push_back(jd);
            }
            all_skinned = all_skinned && has_j;
        }
        if (desc.vertices.empty()) { mesh_ids.push_back(-1); continue; }
        // tangents: taken from the document when every primitive has them, else one per triangle from the uv derivatives (first edge
        // when the uvs are degenerate) with handedness +1 — the frame the shading code builds its bitangent from must never be zero
        {
            bool have_all = any_tangents;
            for (const rfw_vec4& tg : desc.tangents) have_all = have_all && (tg.x != 0.0f || tg.y != 0.0f || tg.z != 0.0f);
            if (!have_all) {
                desc.tangents.assign(desc.vertices.size(), rfw_vec4{1, 0, 0, 1});
                for (size_t t3 = 0; t3 + 2 < desc.vertices.size(); t3 += 3) {
                    const rfw_vec4 &a = desc.vertices[t3], &b = desc.vertices[t3 + 1], &c = desc.vertices[t3 + 2];
                    const float e1[3] = {b.x - a.x, b.y - a.y, b.z - a.z}, e2[3] = {c.x - a.x, c.y - a.y, c.z - a.z};
                    const float du1 = desc.uvs[t3 + 1].x - desc.uvs[t3].x, dv1 = desc.uvs[t3 + 1].y - desc.uvs[t3].y;
                    const float du2 = desc.uvs[t3 + 2].x - desc.uvs[t3].x, dv2 = desc.uvs[t3 + 2].y - desc.uvs[t3].y;
                    const float det = du1 * dv2 - du2 * dv1;
                    float tv[3];
                    for (int k = 0; k < 3; k++) tv[k] = std::fabs(det) > 1e-12f ? (e1[k] * dv2 - e2[k] * dv1) * (1.0f / det) : e1[k];
                    float len2 = tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2];
                    if (!(len2 > 1e-24f)) { for (int k = 0; k < 3; k++) tv[k] = e1[k]; len2 = tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]; }
                    if (!(len2 > 1e-24f)) { tv[0] = 1.0f; tv[1] = tv[2] = 0.0f; len2 = 1.0f; }
                    const float il = 1.0f / std::sqrt(len2);
                    for (size_t k = 0; k < 3; k++) desc.tangents[t3 + k] = rfw_vec4{tv[0] * il, tv[1] * il, tv[2] * il, 1.0f};
                }
            }
        }
        if (!any_normals) desc.normals.assign(desc.vertices.size(), rfw_vec3{0, 0, 0}); // Mesh3D::from generates them (objects_3d/mod.rs:680-711)
        else {
            // a primitive without normals inside a mesh that has some: zero the first normal so the whole mesh is regenerated consistently
            bool missing = false;
            for (const rfw_vec3& nrm : desc.normals) missing = missing || (nrm.x == 0.0f && nrm.y == 0.0f && nrm.z == 0.0f);
            if (missing) desc.normals.assign(desc.vertices.size(), rfw_vec3{0, 0, 0});
        }
        Mesh3D mesh = Mesh3D::from(desc);
        if (all_skinned && joints.size() == mesh.vertices.size()) mesh.skin_data = joints;
        mesh_ids.push_back((int64_t)scene.add_mesh(mesh));
    }

    // ---- node hierarchy -> world matrices
    const std::vector<Json>& nodes = arr("nodes");
    std::vector<M4> world(nodes.size(), m4_identity());
    std::vector<char> visited(nodes.size(), 0);
    auto local_of = [&](const Json& nd) {
        if (const Json* mj = nd.get("matrix")) {
            M4 r = m4_identity();
            for (size_t k = 0; k < 16 && k < mj->size(); k++) r.m[k] = mj->arr[k].num;
            return r;
        }
        double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        if (const Json* j = nd.get("translation")) for (size_t k = 0; k < 3 && k < j->size(); k++) t[k] = j->arr[k].num;
        if (const Json* j = nd.get("rotation")) for (size_t k = 0; k < 4 && k < j->size(); k++) q[k] = j->arr[k].num;
        if (const Json* j = nd.get("scale")) for (size_t k = 0; k < 3 && k < j->size(); k++) s[k] = j->arr[k].num;
        return m4_trs(t, q, s);
    };
    std::vector<std::pair<size_t, M4>> stack;
    std::vector<size_t> order; // nodes reachable from the scene, parents first
    {
        std::vector<size_t> roots;
        const std::vector<Json>& scenes = arr("scenes");
        const size_t si = (size_t)root.integer("scene", 0);
        if (si < scenes.size()) {
            if (const Json* rn = scenes[si].get("nodes"))
                for (const Json& j : rn->arr)
                    if (j.kind == Json::Num && j.num >= 0 && (size_t)j.num < nodes.size()) roots.push_back((size_t)j.num);
        } else {
            std::vector<char> is_child(nodes.size(), 0);
            for (const Json& nd : nodes)
                if (const Json* ch = nd.get("children"))
                    for (const Json& j : ch->arr)
                        if (j.kind == Json::Num && j.num >= 0 && (size_t)j.num < nodes.size()) is_child[(size_t)j.num] = 1;
            for (size_t i = 0; i < nodes.size(); i++)
                if (!is_child[i]) roots.push_back(i);
        }
        for (auto it = roots.rbegin(); it != roots.rend(); ++it) stack.emplace_back(*it, m4_identity());
    }
    while (!stack.empty()) {
        const auto [ni, parent] = stack.back();
        stack.pop_back();
        if (visited[ni]) continue; // a node graph with a cycle or a shared child is invalid glTF: visit once
        visited[ni] = 1;
        world[ni] = m4_mul(parent, local_of(nodes[ni]));
        order.push_back(ni);
        if (const Json* ch = nodes[ni].get("children"))
            for (auto it = ch->arr.rbegin(); it != ch->arr.rend(); ++it)
                if (it->kind == Json::Num && it->num >= 0 && (size_t)it->num < nodes.size()) stack.emplace_back((size_t)it->num, world[ni]);
    }

    // ---- the graph is kept: animations move its nodes later (NodeGraph::set_animation_time)
    NodeGraph graph;
    graph.nodes.resize(nodes.size());
    graph.parent_of.assign(nodes.size(), UINT32_MAX);
    for (const size_t ni : order) graph.order.push_back((uint32_t)ni);
    {
        std::vector<char> is_root(nodes.size(), 1), placed(nodes.size(), 0);
        for (const size_t ni : order) {
            placed[ni] = 1;
            if (const Json* ch = nodes[ni].get("children"))
                for (const Json& j : ch->arr)
                    if (j.kind == Json::Num && j.num >= 0 && (size_t)j.num < nodes.size()) {
                        const size_t c = (size_t)j.num;
                        if (graph.parent_of[c] == UINT32_MAX && !placed[c] && visited[c]) { graph.parent_of[c] = (uint32_t)ni; graph.nodes[ni].children.push_back((uint32_t)c); }
                    }
        }
    }
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        GraphNode& g = graph.nodes[ni];
        const Json& nd = nodes[ni];
        if (const Json* mj = nd.get("matrix")) {
            g.has_matrix = true;
            for (size_t k = 0; k < 16 && k < mj->size(); k++) g.matrix[k] = mj->arr[k].num;
        }
        if (const Json* j = nd.get("translation")) for (size_t k = 0; k < 3 && k < j->size(); k++) g.translation[k] = j->arr[k].num;
        if (const Json* j = nd.get("rotation")) for (size_t k = 0; k < 4 && k < j->size(); k++) g.rotation[k] = j->arr[k].num;
        if (const Json* j = nd.get("scale")) for (size_t k = 0; k < 3 && k < j->size(); k++) g.scale[k] = j->arr[k].num;
        std::memcpy(g.world, world[ni].m, sizeof(g.world));
    }

    // ---- skins: joint_matrices = world(joint) * inverseBind
    std::vector<int32_t> skin_ids;
    for (const Json& sk : arr("skins")) {
        Skin skin;
        const Json* js = sk.get("joints");
        const size_t nj = js ? js->size() : 0;
        std::vector<double> ibm;
        size_t n_ibm = 0;
        if (sk.has("inverseBindMatrices") && !doc.read_accessor(sk.integer("inverseBindMatrices", -1), 16, false, ibm, n_ibm)) { err = doc.err; return false; }
        for (size_t j = 0; j < nj; j++) {
            M4 inv_bind = m4_identity();
            if (j < n_ibm) std::memcpy(inv_bind.m, ibm.data() + 16 * j, sizeof(inv_bind.m));
            const double jn = js->arr[j].kind == Json::Num ? js->arr[j].num : -1.0;
            const M4 jw = (jn >= 0 && (size_t)jn < nodes.size()) ? world[(size_t)jn] : m4_identity();
            skin.inverse_bind_matrices.push_back(to_f32(inv_bind));
            skin.joint_matrices.push_back(to_f32(m4_mul(jw, inv_bind)));
        }
        std::vector<int64_t> joint_nodes;
        for (size_t j = 0; j < nj; j++) {
            const double jn = js->arr[j].kind == Json::Num ? js->arr[j].num : -1.0;
            joint_nodes.push_back((jn >= 0 && (size_t)jn < nodes.size()) ? (int64_t)jn : -1);
        }
        graph.skins.emplace_back((int32_t)scene.skins.size(), std::move(joint_nodes));
        skin_ids.push_back((int32_t)scene.skins.size());
        scene.skins.push_back(std::move(skin));
        scene.skins_changed = true;
    }

    // the node's -Z axis in world space; an axis that is a unit vector to float precision is taken as it is (no second rounding)
    auto minus_z = [](const M4& w, float out[3]) {
        const double dz[3] = {-w.m[8], -w.m[9], -w.m[10]};
        const double len = std::sqrt(dz[0] * dz[0] + dz[1] * dz[1] + dz[2] * dz[2]);
        if (!(len > 0.0)) return;
        const double scale = std::fabs(len - 1.0) < 1e-6 ? 1.0 : 1.0 / len;
        for (int k = 0; k < 3; k++) out[k] = (float)(dz[k] * scale + 0.0); // + 0.0: no negative zeros out of the matrix product
    };
    // ---- KHR_lights_punctual: directional / point / spot lights on nodes (direction = the node's -Z)
    const Json* light_defs = nullptr;
    if (const Json* ext = root.get("extensions"))
        if (const Json* lp = ext->get("KHR_lights_punctual")) light_defs = lp->get("lights");
    auto add_light = [&](const Json& nd, const M4& w) {
        const Json* ext = nd.get("extensions");
        const Json* lp = ext ? ext->get("KHR_lights_punctual") : nullptr;
        if (!lp || !light_defs) return;
        const int64_t li = lp->integer("light", -1);
        if (li < 0 || (size_t)li >= light_defs->size()) return;
        const Json& L = light_defs->arr[(size_t)li];
        const double intensity = L.number("intensity", 1.0);
        double col[3] = {1, 1, 1};
        if (const Json* c = L.get("color"))
            for (size_t k = 0; k < 3 && k < c->size(); k++) col[k] = c->arr[k].num;
        float rad[3] = {(float)(col[0] * intensity), (float)(col[1] * intensity), (float)(col[2] * intensity)};
        if (const Json* ex = L.get("extras"))
            if (const Json* rr = ex->get("rfw_radiance"); rr && rr->size() >= 3)
                for (int k = 0; k < 3; k++) rad[k] = (float)rr->arr[(size_t)k].num;
        const float energy = std::sqrt(rad[0] * rad[0] + rad[1] * rad[1] + rad[2] * rad[2]); // lights.rs: energy = |radiance|
        float dir[3] = {0, 0, -1};
        minus_z(w, dir);
        const rfw_vec3 position{(float)w.m[12], (float)w.m[13], (float)w.m[14]}, radiance{rad[0], rad[1], rad[2]}, direction{dir[0], dir[1], dir[2]};
        const std::string type = L.string("type");
        if (type == "directional") {
            rfw_directional_light l;
            std::memset(&l, 0, sizeof(l));
            l.direction = direction; l.radiance = radiance; l.energy = energy;
            scene.directional_lights.push_back(l);
        } else if (type == "point") {
            rfw_point_light l;
            std::memset(&l, 0, sizeof(l));
            l.position = position; l.radiance = radiance; l.energy = energy;
            scene.point_lights.push_back(l);
        } else if (type == "spot") {
            rfw_spot_light l;
            std::memset(&l, 0, sizeof(l));
            double inner = 0.0, outer = 0.7853981633974483;
            const Json* sp = L.get("spot");
            if (sp) { inner = sp->number("innerConeAngle", 0.0); outer = sp->number("outerConeAngle", 0.7853981633974483); }
            l.position = position; l.radiance = radiance; l.direction = direction; l.energy = energy;
            l.cos_inner = (float)std::cos(inner); l.cos_outer = (float)std::cos(outer);
            if (sp)
                if (const Json* ex = sp->get("extras"))
                    if (const Json* cs = ex->get("rfw_cos"); cs && cs->size() >= 2) { l.cos_inner = (float)cs->arr[0].num; l.cos_outer = (float)cs->arr[1].num; }
            scene.spot_lights.push_back(l);
        }
    };
    // ---- instances, camera, lights
    bool cam_set = false;
    for (const size_t ni : order) {
        const Json& nd = nodes[ni];
        add_light(nd, world[ni]);
        const int64_t mi = nd.integer("mesh", -1);
        if (mi >= 0 && (size_t)mi < mesh_ids.size() && mesh_ids[(size_t)mi] >= 0) {
            const uint32_t mesh = (uint32_t)mesh_ids[(size_t)mi];
            const int64_t si = nd.integer("skin", -1);
            const bool skinned = si >= 0 && (size_t)si < skin_ids.size() && !scene.meshes_3d[mesh].skin_data.empty();
            const size_t slot = scene.add_instance(mesh, skinned ? mat4_identity() : to_f32(world[ni]));
            if (skinned) {
                InstanceList3D& l = scene.instances_3d[mesh];
                if (l.skin_ids.size() <= slot) l.skin_ids.resize(slot + 1, -1);
                l.skin_ids[slot] = skin_ids[(size_t)si];
            }
            graph.nodes[ni].mesh = (int64_t)mesh;
            graph.nodes[ni].instance = (int64_t)slot;
            graph.nodes[ni].skin = skinned ? skin_ids[(size_t)si] : -1;
        }
        const int64_t ci = nd.integer("camera", -1);
        if (cam && !cam_set && ci >= 0 && (size_t)ci < arr("cameras").size()) {
            const Json& cj = arr("cameras")[(size_t)ci];
            if (const Json* persp = cj.get("perspective")) {
                const M4& w = world[ni];
                cam->pos[0] = (float)w.m[12]; cam->pos[1] = (float)w.m[13]; cam->pos[2] = (float)w.m[14];
                minus_z(w, cam->direction); // a glTF camera looks down its local -Z
                cam->fov = (float)(persp->number("yfov", 0.6981317) * 180.0 / 3.14159265358979323846);
                if (persp->has("aspectRatio")) cam->aspect_ratio = (float)persp->number("aspectRatio", 1.0);
                cam->aperture = 0.0f;
                if (const Json* ex = cj.get("extras"))
                    if (const Json* fa = ex->get("rfw_fov_aperture"); fa && fa->size() >= 2) { // written by save_glb: degrees and lens size as floats
                        cam->fov = (float)fa->arr[0].num;
                        cam->aperture = (float)fa->arr[1].num;
                    }
                cam_set = true;
            }
        }
    }
    // ---- animations: channels (node, path) + samplers (key times, values, interpolation)
    for (const Json& an : arr("animations")) {
        Animation anim;
        anim.name = an.string("name");
        const Json* sj = an.get("samplers");
        const Json* cj = an.get("channels");
        std::vector<int> sampler_comps;
        for (size_t i = 0; sj && i < sj->size(); i++) {
            const Json& sm = sj->arr[i];
            AnimationSampler smp;
            const std::string ip = sm.string("interpolation");
            smp.interpolation = ip == "STEP" ? 1 : (ip == "CUBICSPLINE" ? 2 : 0);
            size_t nk = 0, nv = 0;
            if (!doc.read_accessor(sm.integer("input", -1), 1, false, smp.times, nk)) { err = doc.err; return false; }
            const Json* accs = root.get("accessors");
            const int64_t oi = sm.integer("output", -1);
            const int nc = (accs && oi >= 0 && (size_t)oi < accs->size()) ? Doc::components(accs->arr[(size_t)oi].string("type")) : 0;
            if (!doc.read_accessor(oi, 0, true, smp.values, nv)) { err = doc.err; return false; } // normalised integers -> [-1, 1] / [0, 1]
            for (size_t k = 1; k < nk; k++)
                if (!(smp.times[k] >= smp.times[k - 1])) { err = "gltf: animation key times must not decrease"; return false; }
            if (nv != nk * (smp.interpolation == 2 ? 3u : 1u)) { err = "gltf: animation sampler output does not match its key times"; return false; }
            if (nk) anim.duration = std::max(anim.duration, smp.times.back());
            sampler_comps.push_back(nc);
            anim.samplers.push_back(std::move(smp));
        }
        for (size_t i = 0; cj && i < cj->size(); i++) {
            const Json& ch = cj->arr[i];
            const Json* tg = ch.get("target");
            if (!tg) continue;
            const int64_t node = tg->integer("node", -1), smp = ch.integer("sampler", -1);
            const std::string path = tg->string("path");
            const int pi = path == "translation" ? 0 : (path == "rotation" ? 1 : (path == "scale" ? 2 : -1));
            if (pi < 0 || node < 0 || (size_t)node >= nodes.size() || smp < 0 || (size_t)smp >= anim.samplers.size()) continue; // weights, or a target of an extension
            if (sampler_comps[(size_t)smp] != (pi == 1 ? 4 : 3)) { err = "gltf: animation output has the wrong type for its path"; return false; }
            AnimationChannel c;
            c.node = (uint32_t)node; c.sampler = (uint32_t)smp; c.path = pi;
            anim.channels.push_back(c);
        }
        graph.animations.push_back(std::move(anim));
    }
    scene.graphs.push_back(std::move(graph));
    scene.update_lights(); // emissive triangles -> area lights (crates/rfw-scene/src/lib.rs:575-648)
    return true;
}

} // namespace rfw
