// obj.cpp — Wavefront OBJ + MTL -> one rfw mesh and its materials: the reference's ObjLoader (crates/rfw-scene/src/loaders/obj.rs:26-252).
//
// obj.rs hands the parsing to the un-vendored crate tobj 3.0 with `single_index, triangulate, ignore_points, ignore_lines` and then does the
// part restated here line by line: every model of the file is appended, de-indexed, to ONE mesh (:196-237); materials follow :48-190 —
// colour = Kd, specular = Ks, roughness = clamp(1 - log10(Ns) / 1000, 0, 1), transmission = 1 - d, eta = Ni, an emitter's Ke (scaled by
// 10 when every component is <= 1) replaces the colour where it is larger, map_Kd = base colour, norm / map_Ns / map_bump / bump = normal
// map, map_Ke = emissive map, Ps / map_Ps = sheen map, map_Pr / Pm / map_Pm = the metallic-roughness slot; a file without materials gets
// one red material (:188-195).  Images are flipped vertically on load (Flip::FlipV: OBJ's v runs upwards) and shared by path
// (get_texture_index).  tobj's part follows the OBJ format: v / vn / vt, faces `a`, `a/b`, `a//c`, `a/b/c` with negative (relative)
// indices, fan triangulation of polygons, mtllib / usemtl; points and lines are skipped.  The reference ships the material libraries of
// its usual scenes (assets/models/{cbox,sponza/sponza,sibenik/sibenik}.mtl with their textures) but not the geometry (.gitignore:24).
// TGA (the format of assets/models/sponza/textures) is decoded here too: true-colour and grey, plain or run-length encoded.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "rfw_host.hpp"

namespace rfw {

bool decode_tga(const uint8_t* d, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err)
{
    auto fail = [&](const char* m) { err = m; return false; };
    if (size < 18) return fail("tga: truncated header");
    const uint8_t id_len = d[0], cmap_type = d[1], type = d[2], bpp = d[16], desc = d[17];
    const uint32_t cmap_len = (uint32_t)d[5] | ((uint32_t)d[6] << 8), cmap_bits = d[7];
    const uint32_t w = (uint32_t)d[12] | ((uint32_t)d[13] << 8), h = (uint32_t)d[14] | ((uint32_t)d[15] << 8);
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (!(type == 2 || type == 3 || type == 10 || type == 11)) return fail("tga: only true-colour and grey images are supported");
    if (cmap_type > 1 || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 26)) return fail("tga: bad header");
    if (!(grey ? bpp == 8 : (bpp == 24 || bpp == 32))) return fail("tga: unsupported pixel depth");
    const size_t bytes = bpp / 8;
    size_t pos = 18 + (size_t)id_len + (cmap_type ? (size_t)cmap_len * ((cmap_bits + 7) / 8) : 0);
    const size_t npx = (size_t)w * h;
    std::vector<uint8_t> raw(npx * bytes);
    if (!rle) {
        if (pos + raw.size() > size) return fail("tga: truncated pixel data");
        std::memcpy(raw.data(), d + pos, raw.size());
    } else {
        size_t out = 0;
        while (out < npx) {
            if (pos >= size) return fail("tga: truncated run-length data");
            const uint8_t head = d[pos++];
            const size_t count = (size_t)(head & 127) + 1;
            if (out + count > npx) return fail("tga: run past the end of the image");
            if (head & 128) {
                if (pos + bytes > size) return fail("tga: truncated run-length data");
                for (size_t k = 0; k < count; k++) std::memcpy(raw.data() + (out + k) * bytes, d + pos, bytes);
                pos += bytes;
            } else {
                if (pos + count * bytes > size) return fail("tga: truncated run-length data");
                std::memcpy(raw.data() + out * bytes, d + pos, count * bytes);
                pos += count * bytes;
            }
            out += count;
        }
    }
    rgba.resize(npx * 4);
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const uint32_t sy = top_down ? y : h - 1 - y, sx = right_left ? w - 1 - x : x; // stored bottom-up unless bit 5 says otherwise
            const uint8_t* p = raw.data() + ((size_t)sy * w + sx) * bytes;
            uint8_t* o = rgba.data() + ((size_t)y * w + x) * 4;
            if (grey) { o[0] = o[1] = o[2] = p[0]; o[3] = 255; }
            else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = bytes == 4 ? p[3] : 255; } // stored B, G, R (, A)
        }
    width = w;
    height = h;
    return true;
}

namespace {

bool read_all(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

std::string dir_of(const std::string& path)
{
    const size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

std::string lower(std::string s)
{
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

std::string trim(const std::string& s)
{
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// a texture named by a material library: read, decode (PNG / JPEG / TGA), flip vertically, BGRA8 + 5 mips; one scene texture per path
int texture_of(Scene& scene, std::map<std::string, int>& cache, const std::string& dir, std::string name)
{
    name = trim(name);
    if (name.empty()) return -1;
    // options such as "-bm 0.5 file.png" precede the file name: the name is the last token
    if (name[0] == '-') { const size_t sp = name.find_last_of(" \t"); if (sp != std::string::npos) name = name.substr(sp + 1); }
    std::replace(name.begin(), name.end(), '\\', '/');
    if (name.find("..") != std::string::npos || name[0] == '/') return -1; // stays below the document's directory
    const std::string path = dir + name;
    auto it = cache.find(path);
    if (it != cache.end()) return it->second;
    int id = -1;
    std::vector<uint8_t> file, rgba;
    uint32_t w = 0, h = 0;
    std::string err;
    if (read_all(path, file)) {
        const std::string ext = lower(name.size() >= 4 ? name.substr(name.size() - 4) : std::string());
        const bool ok = ext == ".tga" ? decode_tga(file.data(), file.size(), w, h, rgba, err) : decode_image(file.data(), file.size(), w, h, rgba, err);
        if (ok) {
            Texture t;
            t.width = w; t.height = h; t.format = RFW_FORMAT_BGRA8;
            t.bytes.resize(rgba.size());
            for (uint32_t y = 0; y < h; y++) { // Flip::FlipV
                const uint8_t* src = rgba.data() + (size_t)(h - 1 - y) * w * 4;
                uint8_t* dst = t.bytes.data() + (size_t)y * w * 4;
                for (uint32_t x = 0; x < w; x++) { dst[4 * x] = src[4 * x + 2]; dst[4 * x + 1] = src[4 * x + 1]; dst[4 * x + 2] = src[4 * x]; dst[4 * x + 3] = src[4 * x + 3]; }
            }
            t.generate_mipmaps(5);
            id = (int)scene.textures.size();
            scene.textures.push_back(std::move(t));
            scene.textures_changed = true;
        }
    }
    cache[path] = id; // unreadable: -1 (get_texture_index(...).unwrap_or(-1), material/list.rs:431-433)
    return id;
}

struct MtlEntry {
    std::string name;
    float kd[3] = {0, 0, 0}, ks[3] = {0, 0, 0}; // tobj 3.0's Material::default(): black
    float ns = 0.0f, ni = 1.0f, d = 1.0f;
    std::string map_kd, map_normal;
    std::vector<std::pair<std::string, std::string>> unknown; // tobj's unknown_param, in file order
};

bool parse_mtl(const std::string& path, std::vector<MtlEntry>& out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    MtlEntry* cur = nullptr;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        const size_t sp = line.find_first_of(" \t");
        const std::string key = line.substr(0, sp), rest = sp == std::string::npos ? std::string() : trim(line.substr(sp));
        if (key == "newmtl") { out.emplace_back(); cur = &out.back(); cur->name = rest; continue; }
        if (!cur) continue;
        auto f3 = [&](float* v) { std::sscanf(rest.c_str(), "%f %f %f", v, v + 1, v + 2); };
        if (key == "Kd") f3(cur->kd);
        else if (key == "Ks") f3(cur->ks);
        else if (key == "Ka" || key == "illum" || key == "map_Ka" || key == "map_Ks" || key == "map_d" || key == "Tf") {} // parsed by tobj, unused by obj.rs
        else if (key == "Ns") cur->ns = (float)std::atof(rest.c_str());
        else if (key == "Ni") cur->ni = (float)std::atof(rest.c_str());
        else if (key == "d") cur->d = (float)std::atof(rest.c_str());
        else if (key == "map_Kd") cur->map_kd = rest;
        else if (key == "map_Bump" || key == "map_bump" || key == "bump") cur->map_normal = rest;
        else cur->unknown.emplace_back(key, rest);
    }
    return true;
}

} // namespace

bool load_obj(const std::string& path, Scene& scene, std::string& err, bool add_instance, uint32_t* mesh_out)
{
    std::ifstream f(path);
    if (!f) { err = "obj: cannot read " + path; return false; }
    const std::string dir = dir_of(path);
    std::vector<rfw_vec3> pos, nor;
    std::vector<rfw_vec2> tex;
    std::vector<MtlEntry> mtl;
    std::map<std::string, int> mtl_index; // name -> index into mtl (a name defined twice: the later definition, as a map insert overwrites)
    MeshDescriptor desc;
    std::vector<int> tri_material; // index into mtl per triangle, -1: none
    int cur_mat = -1;
    std::string line;
    size_t line_no = 0;
    while (std::getline(f, line)) {
        line_no++;
        // a trailing backslash continues the line
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        while (!line.empty() && line.back() == '\\') {
            line.pop_back();
            std::string more;
            if (!std::getline(f, more)) break;
            line_no++;
            line += " " + more;
        }
        const std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        const size_t sp = t.find_first_of(" \t");
        const std::string key = t.substr(0, sp), rest = sp == std::string::npos ? std::string() : trim(t.substr(sp));
        if (key == "v") {
            rfw_vec3 v{0, 0, 0};
            if (std::sscanf(rest.c_str(), "%f %f %f", &v.x, &v.y, &v.z) != 3) { err = "obj: bad vertex at line " + std::to_string(line_no); return false; }
            pos.push_back(v);
        } else if (key == "vn") {
            rfw_vec3 v{0, 0, 0};
            if (std::sscanf(rest.c_str(), "%f %f %f", &v.x, &v.y, &v.z) != 3) { err = "obj: bad normal at line " + std::to_string(line_no); return false; }
            nor.push_back(v);
        } else if (key == "vt") {
            rfw_vec2 v{0, 0};
            if (std::sscanf(rest.c_str(), "%f %f", &v.x, &v.y) < 1) { err = "obj: bad texture coordinate at line " + std::to_string(line_no); return false; }
            tex.push_back(v);
        } else if (key == "f") {
            struct Corner { long v, t, n; };
            std::vector<Corner> cs;
            std::istringstream ss(rest);
            std::string tok;
            while (ss >> tok) {
                Corner c{0, 0, 0};
                const size_t s1 = tok.find('/');
                const size_t s2 = s1 == std::string::npos ? std::string::npos : tok.find('/', s1 + 1);
                c.v = std::atol(tok.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    const std::string ts = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!ts.empty()) c.t = std::atol(ts.c_str());
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = std::atol(tok.substr(s2 + 1).c_str());
                }
                // 1-based; negative counts back from the elements read so far; 0 = absent
                auto resolve = [&](long i, size_t n, bool required) -> long {
                    if (i > 0 && (size_t)i <= n) return i - 1;
                    if (i < 0 && (size_t)(-i) <= n) return (long)n + i;
                    return required || i != 0 ? -2 : -1;
                };
                c.v = resolve(c.v, pos.size(), true);
                c.t = resolve(c.t, tex.size(), false);
                c.n = resolve(c.n, nor.size(), false);
                if (c.v < 0 || c.t == -2 || c.n == -2) { err = "obj: face index out of range at line " + std::to_string(line_no); return false; }
                cs.push_back(c);
            }
            if (cs.size() < 3) continue; // points and lines written as faces: ignored
            for (size_t k = 1; k + 1 < cs.size(); k++) { // fan
                for (const Corner& c : {cs[0], cs[k], cs[k + 1]}) {
                    const rfw_vec3 p = pos[(size_t)c.v];
                    desc.vertices.push_back(rfw_vec4{p.x, p.y, p.z, 1.0f});
                    desc.normals.push_back(c.n >= 0 ? nor[(size_t)c.n] : rfw_vec3{0, 0, 0});
                    desc.uvs.push_back(c.t >= 0 ? tex[(size_t)c.t] : rfw_vec2{0, 0});
                }
                tri_material.push_back(cur_mat);
            }
        } else if (key == "usemtl") {
            auto it = mtl_index.find(rest);
            cur_mat = it == mtl_index.end() ? -1 : it->second;
        } else if (key == "mtllib") {
            std::istringstream ss(rest);
            std::string name;
            while (ss >> name) {
                std::replace(name.begin(), name.end(), '\\', '/');
                if (name.find("..") != std::string::npos || name[0] == '/') continue;
                const size_t first = mtl.size();
                if (parse_mtl(dir + name, mtl))
                    for (size_t i = first; i < mtl.size(); i++) mtl_index[mtl[i].name] = (int)i;
            }
        }
        // o, g, s, p, l and everything else: no effect on the one mesh obj.rs builds
    }
    if (desc.vertices.empty()) { err = "obj: no triangles in " + path; return false; }

    // ---- materials (obj.rs:48-190)
    std::map<std::string, int> tex_cache;
    std::vector<uint32_t> material_ids;
    for (const MtlEntry& m : mtl) {
        Material mat;
        float color[3] = {m.kd[0], m.kd[1], m.kd[2]};
        float roughness = 1.0f - std::log10(m.ns) / 1000.0f; // Ns = 0: log10 = -inf, clamped to 1 below (f32::max / min ignore nothing here: +inf)
        roughness = std::fmin(std::fmax(roughness, 0.0f), 1.0f);
        if (std::isnan(roughness)) roughness = 0.0f; // f32::max(NaN, 0.0) = 0.0 (a negative Ns)
        std::string normal_map = m.map_normal, rough_map, metal_map, emissive_map, sheen_map;
        for (const auto& kv : m.unknown) {
            const std::string key = lower(kv.first);
            if (key == "ke") {
                float v[3] = {0, 0, 0};
                std::sscanf(kv.second.c_str(), "%f %f %f", v, v + 1, v + 2);
                const bool zero = v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f, small = v[0] <= 1.0f && v[1] <= 1.0f && v[2] <= 1.0f;
                if (!zero && small) for (float& c : v) c *= 10.0f;
                for (int c = 0; c < 3; c++) color[c] = std::fmax(v[c], color[c]);
            } else if (key == "map_pr") rough_map = kv.second;
            else if (key == "map_ke") emissive_map = kv.second;
            else if (key == "ps" || key == "map_ps") sheen_map = kv.second;
            else if (key == "pm" || key == "map_pm") metal_map = kv.second;
            else if (key == "norm" || key == "map_ns" || key == "map_bump") normal_map = kv.second;
        }
        for (int c = 0; c < 3; c++) { mat.color[c] = color[c]; mat.specular[c] = m.ks[c]; }
        mat.color[3] = 1.0f; mat.specular[3] = 1.0f;
        mat.roughness = roughness;
        mat.transmission = 1.0f - m.d; // "opacity" in obj.rs:55 is handed to add_with_maps as the transmission
        mat.eta = m.ni;
        mat.diffuse_tex = texture_of(scene, tex_cache, dir, m.map_kd);
        mat.normal_tex = texture_of(scene, tex_cache, dir, normal_map);
        // obj.rs merges the two grey maps into the channels of one image (:121-147); nothing on the traced path samples that slot
        // (shade.comp reads the base colour and the normal map only), so the slot simply names the first of the two files
        mat.metallic_roughness_tex = texture_of(scene, tex_cache, dir, !rough_map.empty() ? rough_map : metal_map);
        mat.emissive_tex = texture_of(scene, tex_cache, dir, emissive_map);
        mat.sheen_tex = texture_of(scene, tex_cache, dir, sheen_map);
        material_ids.push_back(scene.add_material(mat));
    }
    if (material_ids.empty()) { // :188-195
        Material mat;
        mat.color[0] = 1.0f; mat.color[1] = 0.0f; mat.color[2] = 0.0f; mat.color[3] = 1.0f;
        mat.roughness = 1.0f;
        mat.specular[0] = mat.specular[1] = mat.specular[2] = 0.0f; mat.specular[3] = 1.0f;
        mat.transmission = 1.0f;
        material_ids.push_back(scene.add_material(mat));
    }
    desc.material_ids.reserve(desc.vertices.size());
    for (size_t t = 0; t < tri_material.size(); t++) {
        const int mi = tri_material[t];
        const uint32_t id = (mi >= 0 && (size_t)mi < material_ids.size()) ? material_ids[(size_t)mi] : material_ids[0];
        for (int k = 0; k < 3; k++) desc.material_ids.push_back((int32_t)id);
    }
    desc.name = path;
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(desc));
    if (mesh_out) *mesh_out = mesh;
    if (add_instance) scene.add_instance(mesh, mat4_identity());
    scene.update_lights();
    return true;
}

} // namespace rfw
