// gltf_export.cpp — writes a host Scene as a binary glTF 2.0 file (.glb), the inverse of gltf.cpp for static scenes.
//
// Why it exists: the configurations of BASELINE.json are glTF scenes, and the reference ships none (.gitignore:24).  The procedural
// scenes of rfw_host.cpp become real glTF files through this writer, and bench.py / the tests can then run the path on what the
// importer reads back (`bench.py --via-gltf`): same triangles, same materials, same lights, through the file format.
//
// What is written (all of it core glTF 2.0 or ratified KHR extensions, so any viewer opens the file):
//   * one mesh per Mesh3D, one primitive per material range (non-indexed TRIANGLES; POSITION, NORMAL, TEXCOORD_0, TANGENT as
//     float accessors into one bufferView per attribute);
//   * one node per instance with its 4x4 matrix;
//   * materials as pbrMetallicRoughness (+ emissiveFactor x KHR_materials_emissive_strength for emitters, KHR_materials_ior /
//     KHR_materials_transmission); the Disney parameters glTF has no slot for ride in `extras.rfw_disney`, which the importer
//     reads back (other readers ignore extras);
//   * directional / point / spot lights as KHR_lights_punctual on nodes whose -Z axis is the light direction;
//   * the camera as a perspective camera on a node whose -Z axis is the viewing direction.
// Not written: skins, textures (the procedural scenes that use them are covered by tests/gltf_util.py documents instead).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rfw_host.hpp"

namespace rfw {
namespace {

struct Writer {
    std::string s;
    void raw(const char* t) { s += t; }
    void num(double v)
    {
        char b[40];
        std::snprintf(b, sizeof(b), "%.17g", v);
        s += b;
    }
    void nums(const float* v, int n)
    {
        s += '[';
        for (int i = 0; i < n; i++) {
            if (i) s += ',';
            num((double)v[i]);
        }
        s += ']';
    }
    void key(const char* k) { s += '"'; s += k; s += "\":"; }
};

void put_u32(std::vector<uint8_t>& o, uint32_t v)
{
    for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i)));
}

// a 4x4 (column-major) whose third column is -dir and whose translation is pos: what a glTF camera or punctual light node needs
void look_matrix(const float pos[3], const float dir[3], float m[16])
{
    const float z[3] = {-dir[0], -dir[1], -dir[2]};
    float up[3] = {0, 1, 0};
    if (std::fabs(z[1]) > 0.999f) { up[0] = 1; up[1] = 0; }
    float x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    const float xl = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (float& c : x) c /= xl;
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    const float out[16] = {x[0], x[1], x[2], 0, y[0], y[1], y[2], 0, z[0], z[1], z[2], 0, pos[0], pos[1], pos[2], 1};
    std::memcpy(m, out, sizeof(out));
}

} // namespace

bool save_glb(const std::string& path, const Scene& scene, const Camera3D* cam, std::string& err)
{
    std::vector<uint8_t> bin;
    Writer views, accessors, meshes, nodes, mats, lights;
    int n_views = 0, n_acc = 0, n_nodes = 0, n_meshes = 0, n_lights = 0;
    std::vector<int> scene_nodes;
    std::map<uint32_t, int> mesh_index;

    auto sep = [](Writer& w, int n) { if (n) w.raw(","); };
    auto add_view = [&](const void* data, size_t bytes, int stride) -> int {
        while (bin.size() % 4) bin.push_back(0);
        const size_t off = bin.size();
        bin.insert(bin.end(), (const uint8_t*)data, (const uint8_t*)data + bytes);
        sep(views, n_views);
        views.raw("{\"buffer\":0,\"byteOffset\":"); views.num((double)off);
        views.raw(",\"byteLength\":"); views.num((double)bytes);
        views.raw(",\"byteStride\":"); views.num(stride);
        views.raw(",\"target\":34962}");
        return n_views++;
    };
    auto add_accessor = [&](int view, size_t first, size_t count, int stride, const char* type, const float* mn, const float* mx, int nc) -> int {
        sep(accessors, n_acc);
        accessors.raw("{\"bufferView\":"); accessors.num(view);
        accessors.raw(",\"byteOffset\":"); accessors.num((double)(first * (size_t)stride));
        accessors.raw(",\"componentType\":5126,\"count\":"); accessors.num((double)count);
        accessors.raw(",\"type\":\""); accessors.raw(type); accessors.raw("\"");
        if (mn) { accessors.raw(",\"min\":"); accessors.nums(mn, nc); accessors.raw(",\"max\":"); accessors.nums(mx, nc); }
        accessors.raw("}");
        return n_acc++;
    };

    // ---- meshes
    for (const auto& [id, m] : scene.meshes_3d) {
        const size_t nv = m.vertices.size();
        if (nv == 0) continue;
        std::vector<float> pos(3 * nv), nor(3 * nv), uv(2 * nv), tan(4 * nv);
        bool any_tangent = false;
        for (size_t i = 0; i < nv; i++) {
            const rfw_vertex_3d& v = m.vertices[i];
            pos[3 * i] = v.vertex.x; pos[3 * i + 1] = v.vertex.y; pos[3 * i + 2] = v.vertex.z;
            nor[3 * i] = v.normal.x; nor[3 * i + 1] = v.normal.y; nor[3 * i + 2] = v.normal.z;
            uv[2 * i] = v.uv.x; uv[2 * i + 1] = v.uv.y;
            tan[4 * i] = v.tangent.x; tan[4 * i + 1] = v.tangent.y; tan[4 * i + 2] = v.tangent.z; tan[4 * i + 3] = v.tangent.w;
            any_tangent = any_tangent || v.tangent.x != 0.0f || v.tangent.y != 0.0f || v.tangent.z != 0.0f;
        }
        const int vp = add_view(pos.data(), pos.size() * 4, 12), vn = add_view(nor.data(), nor.size() * 4, 12);
        const int vu = add_view(uv.data(), uv.size() * 4, 8), vt = any_tangent ? add_view(tan.data(), tan.size() * 4, 16) : -1;
        sep(meshes, n_meshes);
        meshes.raw("{\"name\":\""); meshes.raw(m.name.c_str()); meshes.raw("\",\"primitives\":[");
        int n_prims = 0;
        for (const rfw_vertex_mesh& r : m.ranges) {
            if (r.last <= r.first || r.last > nv) continue;
            const size_t first = r.first, count = r.last - r.first;
            float mn[3] = {pos[3 * first], pos[3 * first + 1], pos[3 * first + 2]}, mx[3] = {mn[0], mn[1], mn[2]};
            for (size_t i = first; i < first + count; i++)
                for (int c = 0; c < 3; c++) { mn[c] = std::fmin(mn[c], pos[3 * i + c]); mx[c] = std::fmax(mx[c], pos[3 * i + c]); }
            const int ap = add_accessor(vp, first, count, 12, "VEC3", mn, mx, 3), an = add_accessor(vn, first, count, 12, "VEC3", nullptr, nullptr, 0);
            const int au = add_accessor(vu, first, count, 8, "VEC2", nullptr, nullptr, 0);
            const int at = any_tangent ? add_accessor(vt, first, count, 16, "VEC4", nullptr, nullptr, 0) : -1;
            sep(meshes, n_prims++);
            meshes.raw("{\"mode\":4,\"material\":"); meshes.num(r.mat_id);
            meshes.raw(",\"attributes\":{\"POSITION\":"); meshes.num(ap);
            meshes.raw(",\"NORMAL\":"); meshes.num(an);
            meshes.raw(",\"TEXCOORD_0\":"); meshes.num(au);
            if (at >= 0) { meshes.raw(",\"TANGENT\":"); meshes.num(at); }
            meshes.raw("}}");
        }
        meshes.raw("]}");
        mesh_index[id] = n_meshes++;
    }
    // ---- instances
    for (const auto& [id, list] : scene.instances_3d) {
        const auto it = mesh_index.find(id);
        if (it == mesh_index.end()) continue;
        for (const rfw_mat4& mtx : list.matrices) {
            bool zero = true; // an invalidated slot (instances_3d.rs:79-86)
            for (int k = 0; k < 16; k++) zero = zero && ((const float*)&mtx)[k] == 0.0f;
            if (zero) continue;
            sep(nodes, n_nodes);
            nodes.raw("{\"mesh\":"); nodes.num(it->second);
            nodes.raw(",\"matrix\":"); nodes.nums((const float*)&mtx, 16);
            nodes.raw("}");
            scene_nodes.push_back(n_nodes++);
        }
    }
    // ---- materials
    int n_mats = 0;
    bool emissive_ext = false, ior_ext = false, tr_ext = false;
    for (const Material& m : scene.materials) {
        sep(mats, n_mats++);
        const bool emitter = is_emissive(m);
        const float black[4] = {0, 0, 0, 1};
        mats.raw("{\"pbrMetallicRoughness\":{\"baseColorFactor\":"); mats.nums(emitter ? black : m.color, 4);
        mats.raw(",\"metallicFactor\":"); mats.num(m.metallic);
        mats.raw(",\"roughnessFactor\":"); mats.num(m.roughness);
        mats.raw("}");
        mats.raw(",\"extensions\":{");
        int n_ext = 0;
        if (emitter) {
            const float strength = std::fmax(m.color[0], std::fmax(m.color[1], m.color[2]));
            mats.raw("\"KHR_materials_emissive_strength\":{\"emissiveStrength\":"); mats.num(strength); mats.raw("}");
            n_ext++;
            emissive_ext = true;
        }
        if (m.eta != 1.0f) { sep(mats, n_ext++); mats.raw("\"KHR_materials_ior\":{\"ior\":"); mats.num(m.eta); mats.raw("}"); ior_ext = true; }
        if (m.transmission != 0.0f) { sep(mats, n_ext++); mats.raw("\"KHR_materials_transmission\":{\"transmissionFactor\":"); mats.num(m.transmission); mats.raw("}"); tr_ext = true; }
        mats.raw("}");
        if (emitter) {
            const double strength = std::fmax(m.color[0], std::fmax(m.color[1], m.color[2]));
            mats.raw(",\"emissiveFactor\":[");
            for (int c = 0; c < 3; c++) { if (c) mats.raw(","); mats.num((double)m.color[c] / strength); }
            mats.raw("]");
        }
        // Disney parameters without a glTF slot, and the exact colour of an emitter (factor x strength rounds)
        mats.raw(",\"extras\":{\"rfw_disney\":{\"color\":"); mats.nums(m.color, 4);
        mats.raw(",\"absorption\":"); mats.nums(m.absorption, 4);
        mats.raw(",\"specular\":"); mats.nums(m.specular, 4);
        const float p[] = {m.subsurface, m.specular_f, m.specular_tint, m.anisotropic, m.sheen, m.sheen_tint, m.clearcoat, m.clearcoat_gloss,
                           m.eta, m.custom0, m.custom1, m.custom2, m.custom3};
        mats.raw(",\"params\":"); mats.nums(p, 13);
        mats.raw("}}}");
    }
    // ---- punctual lights
    auto light_node = [&](const float pos[3], const float dir[3]) {
        float mtx[16];
        look_matrix(pos, dir, mtx);
        sep(nodes, n_nodes);
        nodes.raw("{\"matrix\":"); nodes.nums(mtx, 16);
        nodes.raw(",\"extensions\":{\"KHR_lights_punctual\":{\"light\":"); nodes.num(n_lights); nodes.raw("}}}");
        scene_nodes.push_back(n_nodes++);
    };
    auto light_colour = [&](const rfw_vec3& radiance) {
        const double mx = std::fmax(radiance.x, std::fmax(radiance.y, radiance.z));
        lights.raw("\"color\":["); lights.num(mx > 0 ? radiance.x / mx : 0); lights.raw(","); lights.num(mx > 0 ? radiance.y / mx : 0); lights.raw(",");
        lights.num(mx > 0 ? radiance.z / mx : 0); lights.raw("],\"intensity\":"); lights.num(mx);
        const float exact[3] = {radiance.x, radiance.y, radiance.z};
        lights.raw(",\"extras\":{\"rfw_radiance\":"); lights.nums(exact, 3); lights.raw("}");
    };
    const float origin[3] = {0, 0, 0}, down[3] = {0, -1, 0};
    for (const rfw_directional_light& l : scene.directional_lights) {
        const float d[3] = {l.direction.x, l.direction.y, l.direction.z};
        light_node(origin, d);
        sep(lights, n_lights++);
        lights.raw("{\"type\":\"directional\","); light_colour(l.radiance); lights.raw("}");
    }
    for (const rfw_point_light& l : scene.point_lights) {
        const float p[3] = {l.position.x, l.position.y, l.position.z};
        light_node(p, down);
        sep(lights, n_lights++);
        lights.raw("{\"type\":\"point\","); light_colour(l.radiance); lights.raw("}");
    }
    for (const rfw_spot_light& l : scene.spot_lights) {
        const float p[3] = {l.position.x, l.position.y, l.position.z}, d[3] = {l.direction.x, l.direction.y, l.direction.z};
        light_node(p, d);
        sep(lights, n_lights++);
        lights.raw("{\"type\":\"spot\","); light_colour(l.radiance);
        lights.raw(",\"spot\":{\"innerConeAngle\":"); lights.num(std::acos((double)l.cos_inner));
        lights.raw(",\"outerConeAngle\":"); lights.num(std::acos((double)l.cos_outer));
        const float cs[2] = {l.cos_inner, l.cos_outer};
        lights.raw(",\"extras\":{\"rfw_cos\":"); lights.nums(cs, 2); lights.raw("}}}");
    }
    // ---- camera
    if (cam) {
        float mtx[16];
        look_matrix(cam->pos, cam->direction, mtx);
        sep(nodes, n_nodes);
        nodes.raw("{\"camera\":0,\"matrix\":"); nodes.nums(mtx, 16); nodes.raw("}");
        scene_nodes.push_back(n_nodes++);
    }

    Writer j;
    j.raw("{\"asset\":{\"version\":\"2.0\",\"generator\":\"rfw-rs_amd host\"}");
    if (emissive_ext || ior_ext || tr_ext || n_lights) {
        j.raw(",\"extensionsUsed\":[");
        int n = 0;
        if (emissive_ext) { sep(j, n++); j.raw("\"KHR_materials_emissive_strength\""); }
        if (ior_ext) { sep(j, n++); j.raw("\"KHR_materials_ior\""); }
        if (tr_ext) { sep(j, n++); j.raw("\"KHR_materials_transmission\""); }
        if (n_lights) { sep(j, n++); j.raw("\"KHR_lights_punctual\""); }
        j.raw("]");
    }
    if (n_lights) { j.raw(",\"extensions\":{\"KHR_lights_punctual\":{\"lights\":["); j.s += lights.s; j.raw("]}}"); }
    j.raw(",\"scene\":0,\"scenes\":[{\"nodes\":[");
    for (size_t i = 0; i < scene_nodes.size(); i++) { if (i) j.raw(","); j.num(scene_nodes[i]); }
    j.raw("]}],\"nodes\":["); j.s += nodes.s; j.raw("]");
    if (n_meshes) { j.raw(",\"meshes\":["); j.s += meshes.s; j.raw("]"); }
    if (n_mats) { j.raw(",\"materials\":["); j.s += mats.s; j.raw("]"); }
    if (n_acc) { j.raw(",\"accessors\":["); j.s += accessors.s; j.raw("]"); j.raw(",\"bufferViews\":["); j.s += views.s; j.raw("]"); }
    if (cam) {
        j.raw(",\"cameras\":[{\"type\":\"perspective\",\"perspective\":{\"yfov\":"); j.num((double)cam->fov * 3.14159265358979323846 / 180.0);
        j.raw(",\"aspectRatio\":"); j.num(cam->aspect_ratio);
        j.raw(",\"znear\":"); j.num(cam->near_plane); j.raw(",\"zfar\":"); j.num(cam->far_plane);
        const float ex[2] = {cam->fov, cam->aperture};
        j.raw("},\"extras\":{\"rfw_fov_aperture\":"); j.nums(ex, 2); j.raw("}}]");
    }
    while (bin.size() % 4) bin.push_back(0);
    if (!bin.empty()) { j.raw(",\"buffers\":[{\"byteLength\":"); j.num((double)bin.size()); j.raw("}]"); }
    j.raw("}");
    while (j.s.size() % 4) j.s += ' ';

    const uint64_t total = 12 + 8 + j.s.size() + (bin.empty() ? 0 : 8 + bin.size());
    if (total > 0xffffffffull) { err = "save_glb: scene larger than the 4 GiB a .glb can hold"; return false; }
    std::vector<uint8_t> head;
    put_u32(head, 0x46546C67u); put_u32(head, 2); put_u32(head, (uint32_t)total);
    put_u32(head, (uint32_t)j.s.size()); put_u32(head, 0x4E4F534Au);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "save_glb: cannot open " + path; return false; }
    bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size() && std::fwrite(j.s.data(), 1, j.s.size(), f) == j.s.size();
    if (ok && !bin.empty()) {
        std::vector<uint8_t> bh;
        put_u32(bh, (uint32_t)bin.size()); put_u32(bh, 0x004E4942u);
        ok = std::fwrite(bh.data(), 1, bh.size(), f) == bh.size() && std::fwrite(bin.data(), 1, bin.size(), f) == bin.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = "save_glb: write failed: " + path;
    return ok;
}

} // namespace rfw
