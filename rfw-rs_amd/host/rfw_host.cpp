// rfw_host.cpp — scene-side inputs of the Backend boundary + synthetic scenes (see rfw_host.hpp).
#include "rfw_host.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace rfw {

namespace {
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
inline rfw_vec3 pod(V3 a) { return rfw_vec3{a.x, a.y, a.z}; }
inline V3 unpod(rfw_vec3 a) { return V3{a.x, a.y, a.z}; }

inline void aabb_reset(rfw_aabb& b)
{
    for (int i = 0; i < 3; i++) { b.min[i] = 1e34f; b.max[i] = -1e34f; }
    b.extra1 = 0; b.extra2 = 0;
}
inline void aabb_grow(rfw_aabb& b, V3 p)
{
    const float v[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; i++) { b.min[i] = std::min(b.min[i], v[i]); b.max[i] = std::max(b.max[i], v[i]); }
}
// crates/rfw-backend/src/structs.rs:970-984
inline V3 tri_normal(V3 v0, V3 v1, V3 v2) { return normalize(cross(v1 - v0, v2 - v0)); }
inline float tri_area(V3 v0, V3 v1, V3 v2)
{
    const float a = length(v1 - v0), b = length(v2 - v1), c = length(v0 - v2);
    const float s = (a + b + c) * 0.5f;
    return std::sqrt(s * (s - a) * (s - b) * (s - c));
}
inline V3 mul_point(const rfw_mat4& m, V3 p)
{
    return V3{m.m[0] * p.x + m.m[4] * p.y + m.m[8] * p.z + m.m[12], m.m[1] * p.x + m.m[5] * p.y + m.m[9] * p.z + m.m[13],
              m.m[2] * p.x + m.m[6] * p.y + m.m[10] * p.z + m.m[14]};
}
struct Rng {
    uint32_t s;
    explicit Rng(uint32_t seed) : s(seed * 747796405u + 2891336453u) { next(); }
    uint32_t next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
    float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * uniform(); }
};
} // namespace

// ------------------------------------------------------------------ materials
rfw_device_material into_device_material(const Material& mat)
{
    // `(f * 255.0).min(255.0) as u8` (material/list.rs:756): Rust's float -> integer `as` saturates and maps NaN to 0 (f32::min returns the
    // other operand for a NaN, so a NaN parameter becomes 255.0 first); a C++ cast of a negative float would wrap instead
    auto to_char = [](float f) -> uint32_t {
        float x = f * 255.0f;
        x = (x != x) ? 255.0f : std::min(x, 255.0f);
        return x <= 0.0f ? 0u : (uint32_t)(uint8_t)x;
    };
    auto to_u32 = [&](float a, float b, float c, float d) -> uint32_t { return to_char(a) | (to_char(b) << 8) | (to_char(c) << 16) | (to_char(d) << 24); };
    rfw_device_material d;
    std::memset(&d, 0, sizeof(d));
    std::memcpy(d.color, mat.color, 16);
    std::memcpy(d.absorption, mat.absorption, 16);
    std::memcpy(d.specular, mat.specular, 16);
    d.parameters[0] = to_u32(mat.metallic, mat.subsurface, mat.specular_f, mat.roughness);
    d.parameters[1] = to_u32(mat.specular_tint, mat.anisotropic, mat.sheen, mat.sheen_tint);
    d.parameters[2] = to_u32(mat.clearcoat, mat.clearcoat_gloss, mat.transmission, mat.eta);
    d.parameters[3] = to_u32(mat.custom0, mat.custom1, mat.custom2, mat.custom3);
    uint32_t flags = 0;
    if (mat.diffuse_tex >= 0) flags |= RFW_MAT_HAS_DIFFUSE_MAP;
    if (mat.normal_tex >= 0) flags |= RFW_MAT_HAS_NORMAL_MAP;
    if (mat.metallic_roughness_tex >= 0) flags |= RFW_MAT_HAS_ROUGHNESS_MAP | RFW_MAT_HAS_METALLIC_MAP;
    if (mat.emissive_tex >= 0) flags |= RFW_MAT_HAS_EMISSIVE_MAP;
    if (mat.sheen_tex >= 0) flags |= RFW_MAT_HAS_SHEEN_MAP;
    d.flags = flags;
    d.diffuse_map = mat.diffuse_tex;
    d.normal_map = mat.normal_tex;
    d.metallic_roughness_map = mat.metallic_roughness_tex;
    d.emissive_map = mat.emissive_tex;
    d.sheen_map = mat.sheen_tex;
    return d;
}
bool is_emissive(const Material& m) { return m.color[0] > 1.0f || m.color[1] > 1.0f || m.color[2] > 1.0f; }

// ------------------------------------------------------------------ Mesh3D::from(MeshDescriptor)
Mesh3D Mesh3D::from(const MeshDescriptor& desc)
{
    Mesh3D out;
    out.name = desc.name;
    const size_t nv = desc.vertices.size();
    const size_t nt = nv / 3;
    aabb_reset(out.bounds);
    auto P = [&](size_t i) { return v3(desc.vertices[i].x, desc.vertices[i].y, desc.vertices[i].z); };

    std::vector<uint32_t> material_ids(nt);
    for (size_t i = 0; i < nt; i++) material_ids[i] = (uint32_t)desc.material_ids[3 * i];

    // objects_3d/mod.rs:680-711: generate area-weighted normals when the descriptor has none
    std::vector<V3> normals(nv);
    const bool gen = desc.normals.empty() || (desc.normals[0].x == 0.0f && desc.normals[0].y == 0.0f && desc.normals[0].z == 0.0f);
    if (gen) {
        for (size_t i = 0; i < nv; i += 3) {
            const V3 v0 = P(i), v1 = P(i + 1), v2 = P(i + 2);
            const V3 n = normalize(cross(v1 - v0, v2 - v0)) * tri_area(v0, v1, v2);
            normals[i] = normalize(n);
            normals[i + 1] = normalize(n);
            normals[i + 2] = normalize(n);
        }
    } else {
        for (size_t i = 0; i < nv; i++) normals[i] = unpod(desc.normals[i]);
    }
    for (size_t i = 0; i < nv; i++) aabb_grow(out.bounds, P(i));

    out.vertices.resize(nv);
    for (size_t i = 0; i < nv; i++) {
        rfw_vertex_3d v;
        std::memset(&v, 0, sizeof(v));
        v.vertex = desc.vertices[i];
        v.normal = pod(normals[i]);
        v.mat_id = material_ids[i / 3];
        v.uv = desc.uvs.empty() ? rfw_vec2{0, 0} : desc.uvs[i];
        v.tangent = desc.tangents.empty() ? rfw_vec4{0, 0, 0, 0} : desc.tangents[i];
        out.vertices[i] = v;
    }

    // objects_3d/mod.rs:733-785: one VertexMesh per run of equal material ids (first/last are vertex indices)
    {
        uint32_t start = 0;
        rfw_aabb vb;
        aabb_reset(vb);
        for (size_t i = 0; i < nt; i++) {
            if (i > 0 && material_ids[i] != material_ids[i - 1]) {
                out.ranges.push_back(rfw_vertex_mesh{vb, start * 3, (uint32_t)i * 3, material_ids[i - 1], 0});
                start = (uint32_t)i;
                aabb_reset(vb);
            }
            for (int j = 0; j < 3; j++) aabb_grow(vb, P(3 * i + j));
        }
        if (nt > 0) out.ranges.push_back(rfw_vertex_mesh{vb, start * 3, (uint32_t)nt * 3, material_ids[nt - 1], 0});
    }

    // objects_3d/mod.rs:787-850
    out.triangles.resize(nt);
    for (size_t i = 0; i < nt; i++) {
        const size_t i0 = 3 * i, i1 = i0 + 1, i2 = i0 + 2;
        const V3 v0 = P(i0), v1 = P(i1), v2 = P(i2);
        const rfw_vec2 uv0 = desc.uvs.empty() ? rfw_vec2{0, 0} : desc.uvs[i0];
        const rfw_vec2 uv1 = desc.uvs.empty() ? rfw_vec2{0, 0} : desc.uvs[i1];
        const rfw_vec2 uv2 = desc.uvs.empty() ? rfw_vec2{0, 0} : desc.uvs[i2];
        const float ta = (float)(1024 * 1024) * std::fabs((uv1.x - uv0.x) * (uv2.y - uv0.y) - (uv2.x - uv0.x) * (uv1.y - uv0.y));
        const float pa = length(cross(v1 - v0, v2 - v0));
        float lod = std::sqrt(0.5f * std::log2(ta / pa));
        if (!(lod > 0.0f)) lod = 0.0f; // 0.0_f32.max(NaN) == 0
        rfw_rt_triangle t;
        std::memset(&t, 0, sizeof(t));
        t.vertex0 = pod(v0); t.u0 = uv0.x;
        t.vertex1 = pod(v1); t.u1 = uv1.x;
        t.vertex2 = pod(v2); t.u2 = uv2.x;
        t.normal = pod(tri_normal(v0, v1, v2)); t.v0 = uv0.y;
        t.n0 = pod(normals[i0]); t.v1 = uv1.y;
        t.n1 = pod(normals[i1]); t.v2 = uv2.y;
        t.n2 = pod(normals[i2]); t.id = (int32_t)i;
        if (!desc.tangents.empty()) {
            t.tangent0 = desc.tangents[i0];
            t.tangent1 = desc.tangents[i1];
            t.tangent2 = desc.tangents[i2]; // the reference reads tangents[i1] here (objects_3d/mod.rs:818) — not inherited
        }
        t.light_id = -1;
        t.mat_id = (int32_t)material_ids[i];
        t.lod = lod;
        t.area = tri_area(v0, v1, v2);
        out.triangles[i] = t;
    }
    out.materials = material_ids;
    return out;
}

rfw_mesh_data_3d Mesh3D::as_data() const
{
    rfw_mesh_data_3d d;
    std::memset(&d, 0, sizeof(d));
    d.vertices = vertices.data(); d.num_vertices = (uint32_t)vertices.size();
    d.triangles = triangles.data(); d.num_triangles = (uint32_t)triangles.size();
    d.ranges = ranges.data(); d.num_ranges = (uint32_t)ranges.size();
    d.skin_data = skin_data.empty() ? nullptr : skin_data.data(); d.num_skin_data = (uint32_t)skin_data.size();
    d.flags = flags;
    d.bounds = bounds;
    return d;
}

size_t InstanceList3D::allocate(const rfw_mat4& m)
{
    matrices.push_back(m);
    skin_ids.push_back(-1);
    flags.push_back(RFW_INSTANCE_TRANSFORMED);
    return matrices.size() - 1;
}
void InstanceList3D::make_invalid(size_t slot)
{
    std::memset(&matrices[slot], 0, sizeof(rfw_mat4));
    flags[slot] = RFW_INSTANCE_TRANSFORMED;
}

rfw_texture_data Texture::as_data() const
{
    rfw_texture_data d;
    d.width = width; d.height = height; d.mip_levels = mip_levels; d.bytes = bytes.data(); d.format = format;
    return d;
}
void Texture::generate_mipmaps(uint32_t levels)
{
    bytes.resize((size_t)width * height * 4);
    mip_levels = 1;
    size_t src_off = 0;
    uint32_t w = width, h = height;
    for (uint32_t l = 1; l < levels && (w >> 1) > 0 && (h >> 1) > 0; l++) {
        const uint32_t nw = w >> 1, nh = h >> 1;
        const size_t dst_off = bytes.size();
        bytes.resize(dst_off + (size_t)nw * nh * 4);
        for (uint32_t y = 0; y < nh; y++)
            for (uint32_t x = 0; x < nw; x++)
                for (int c = 0; c < 4; c++) {
                    const uint32_t s = bytes[src_off + ((size_t)(2 * y) * w + 2 * x) * 4 + c] + bytes[src_off + ((size_t)(2 * y) * w + 2 * x + 1) * 4 + c] +
                                       bytes[src_off + ((size_t)(2 * y + 1) * w + 2 * x) * 4 + c] + bytes[src_off + ((size_t)(2 * y + 1) * w + 2 * x + 1) * 4 + c];
                    bytes[dst_off + ((size_t)y * nw + x) * 4 + c] = (uint8_t)((s + 2) / 4);
                }
        src_off = dst_off;
        w = nw; h = nh;
        mip_levels++;
    }
}

// ------------------------------------------------------------------ camera/mod.rs:164-186, frame from calculate_matrix (:246-252)
static void camera_frame(const Camera3D& c, V3& x, V3& y, V3& z)
{
    z = normalize(v3(c.direction[0], c.direction[1], c.direction[2]));
    x = normalize(cross(z, v3(0, 1, 0)));
    y = normalize(cross(x, z));
}
void Camera3D::translate_relative(const float delta[3])
{
    V3 x, y, z;
    camera_frame(*this, x, y, z);
    const V3 d = v3(delta[0], delta[1], delta[2]) * speed;
    const V3 p = v3(pos[0], pos[1], pos[2]) + (d.x * x + d.y * y + d.z * z);
    pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
}
void Camera3D::translate_target(const float delta[3])
{
    V3 x, y, z;
    camera_frame(*this, x, y, z);
    const V3 d = normalize(v3(direction[0], direction[1], direction[2]) + delta[0] * x + delta[1] * y + delta[2] * z);
    direction[0] = d.x; direction[1] = d.y; direction[2] = d.z;
}
void Camera3D::look_at(const float origin[3], const float target[3])
{
    const V3 d = normalize(v3(target[0] - origin[0], target[1] - origin[1], target[2] - origin[2]));
    for (int k = 0; k < 3; k++) pos[k] = origin[k];
    direction[0] = d.x; direction[1] = d.y; direction[2] = d.z;
}

// ------------------------------------------------------------------ Camera3D::get_view (camera/mod.rs:77-115, 246-252)
rfw_camera_view_3d Camera3D::get_view(uint32_t width, uint32_t height) const
{
    const float PI = 3.14159265358979323846f;
    const V3 yup = v3(0, 1, 0);
    const V3 z = normalize(v3(direction[0], direction[1], direction[2]));
    const V3 x = normalize(cross(z, yup));
    const V3 y = normalize(cross(x, z));
    const V3 right = x, up = y, forward = z;
    const V3 p = v3(pos[0], pos[1], pos[2]);
    const float spread_angle = (fov * PI / 180.0f) * (1.0f / (float)height);
    const float screen_size = std::tan(fov * 0.5f / (180.0f / PI));
    const V3 center = p + focal_distance * forward;
    const V3 p1 = center - screen_size * right * focal_distance * aspect_ratio + screen_size * focal_distance * up;
    const V3 p2 = center + screen_size * right * focal_distance * aspect_ratio + screen_size * focal_distance * up;
    const V3 p3 = center - screen_size * right * focal_distance * aspect_ratio - screen_size * focal_distance * up;
    rfw_camera_view_3d v;
    std::memset(&v, 0, sizeof(v));
    v.pos = pod(p);
    v.lens_size = aperture;
    v.right = pod(p2 - p1);
    v.p1 = pod(p1);
    v.direction = pod(forward);
    v.spread_angle = spread_angle;
    v.up = pod(p3 - p1);
    v.epsilon = 1e-4f; // crates/rfw-scene/src/constants.rs:1
    v.inv_width = 1.0f / (float)width;
    v.inv_height = 1.0f / (float)height;
    v.aspect_ratio = aspect_ratio;
    v.fov = fov * PI / 180.0f;
    v.near_plane = near_plane;
    v.far_plane = far_plane;
    return v;
}

// ------------------------------------------------------------------ Scene
uint32_t Scene::add_material(const Material& m)
{
    materials.push_back(m);
    material_changed_bits.clear(); // the list grew: everything is handed over again
    materials_changed = true;
    return (uint32_t)materials.size() - 1;
}
uint32_t Scene::add_mesh(const Mesh3D& m)
{
    const uint32_t id = meshes_3d.empty() ? 0u : meshes_3d.rbegin()->first + 1;
    meshes_3d[id] = m;
    instances_3d[id];
    mesh_changed[id] = true;
    return id;
}
void Scene::replace_mesh(uint32_t id, const Mesh3D& m)
{
    meshes_3d[id] = m;
    mesh_changed[id] = true;
    instances_changed[id] = true; // the local bounds travel with the instance list
}
void Scene::remove_mesh(uint32_t id)
{
    if (!meshes_3d.erase(id)) return;
    instances_3d.erase(id);
    mesh_changed.erase(id);
    instances_changed.erase(id);
    removed_meshes.push_back(id);
}
void Scene::set_material(uint32_t index, const Material& m)
{
    if (index >= materials.size()) return;
    materials[index] = m;
    if (!materials_changed) material_changed_bits.assign((materials.size() + 31) / 32, 0u); // first edit since the last synchronize
    if (!material_changed_bits.empty()) material_changed_bits[index / 32] |= 1u << (index % 32);
    materials_changed = true;
}
void Scene::set_texture(uint32_t index, const Texture& t)
{
    if (index >= textures.size()) return;
    textures[index] = t;
    if (!textures_changed) texture_changed_bits.assign((textures.size() + 31) / 32, 0u); // first edit since the last synchronize
    if (!texture_changed_bits.empty()) texture_changed_bits[index / 32] |= 1u << (index % 32);
    textures_changed = true;
}
size_t Scene::add_instance(uint32_t mesh, const rfw_mat4& m)
{
    instances_changed[mesh] = true;
    return instances_3d[mesh].allocate(m);
}
void Scene::set_matrix(uint32_t mesh, size_t slot, const rfw_mat4& m)
{
    instances_3d[mesh].matrices[slot] = m;
    instances_3d[mesh].flags[slot] |= RFW_INSTANCE_TRANSFORMED;
    instances_changed[mesh] = true;
}
std::vector<rfw_device_material> Scene::device_materials() const
{
    std::vector<rfw_device_material> d(materials.size());
    for (size_t i = 0; i < materials.size(); i++) d[i] = into_device_material(materials[i]);
    return d;
}
uint64_t Scene::triangle_count() const
{
    uint64_t n = 0;
    for (auto& kv : meshes_3d) {
        auto it = instances_3d.find(kv.first);
        const uint64_t k = it == instances_3d.end() ? 0 : it->second.matrices.size();
        n += (uint64_t)kv.second.triangles.size() * k;
    }
    return n;
}

// crates/rfw-scene/src/lib.rs:575-648: one AreaLight per emissive triangle per instance; light ids written back to the
// mesh triangles.  Global instance numbering: mesh_base[mesh] + slot.
void Scene::update_lights()
{
    area_lights.clear();
    int32_t base = 0;
    for (auto& kv : meshes_3d) {
        Mesh3D& m = kv.second;
        InstanceList3D& insts = instances_3d[kv.first];
        for (size_t s = 0; s < insts.matrices.size(); s++) {
            const rfw_mat4& transform = insts.matrices[s];
            bool zero = true;
            for (int i = 0; i < 16; i++) zero = zero && transform.m[i] == 0.0f;
            if (zero) continue;
            for (const rfw_vertex_mesh& r : m.ranges) {
                if (r.mat_id >= materials.size() || !is_emissive(materials[r.mat_id])) continue;
                // NOTE: the reference iterates triangle i but reads vertices[i], [i+1], [i+2] (lib.rs:601-607) — an
                // indexing bug (vertex index, not 3*i); not inherited: triangle i is vertices 3i..3i+2.
                for (uint32_t i = r.first / 3; i < r.last / 3; i++) {
                    const V3 v0 = mul_point(transform, v3(m.vertices[3 * i].vertex.x, m.vertices[3 * i].vertex.y, m.vertices[3 * i].vertex.z));
                    const V3 v1 = mul_point(transform, v3(m.vertices[3 * i + 1].vertex.x, m.vertices[3 * i + 1].vertex.y, m.vertices[3 * i + 1].vertex.z));
                    const V3 v2 = mul_point(transform, v3(m.vertices[3 * i + 2].vertex.x, m.vertices[3 * i + 2].vertex.y, m.vertices[3 * i + 2].vertex.z));
                    const Material& mat = materials[r.mat_id];
                    // AreaLight::new (crates/rfw-backend/src/lights.rs:70-97)
                    rfw_area_light al;
                    std::memset(&al, 0, sizeof(al));
                    const V3 radiance = v3(std::fabs(mat.color[0]), std::fabs(mat.color[1]), std::fabs(mat.color[2]));
                    al.position = pod((v0 + v1 + v2) * (1.0f / 3.0f));
                    al.energy = length(radiance);
                    al.normal = pod(tri_normal(v0, v1, v2));
                    al.area = tri_area(v0, v1, v2);
                    al.vertex0 = pod(v0); al.inst_idx = base + (int32_t)s;
                    al.vertex1 = pod(v1); al.mesh_id = (int32_t)kv.first;
                    al.radiance = pod(radiance); al._dummy1 = 1;
                    al.vertex2 = pod(v2); al._dummy2 = 2;
                    m.triangles[i].light_id = (int32_t)area_lights.size();
                    area_lights.push_back(al);
                    mesh_changed[kv.first] = true;
                }
            }
        }
        base += (int32_t)insts.matrices.size();
    }
    lights_changed = true;
}

rfw_mat4 mat4_identity()
{
    rfw_mat4 m;
    std::memset(&m, 0, sizeof(m));
    m.m[0] = m.m[5] = m.m[10] = m.m[15] = 1.0f;
    return m;
}
rfw_mat4 mat4_from_translation(float x, float y, float z)
{
    rfw_mat4 m = mat4_identity();
    m.m[12] = x; m.m[13] = y; m.m[14] = z;
    return m;
}
rfw_mat4 mat4_from_scale_translation(float s, float x, float y, float z)
{
    rfw_mat4 m = mat4_from_translation(x, y, z);
    m.m[0] = m.m[5] = m.m[10] = s;
    return m;
}
rfw_mat4 mat4_from_trs(const float t[3], const float axis_in[3], float angle, float scale)
{
    const V3 a = normalize(v3(axis_in[0], axis_in[1], axis_in[2]));
    const float c = std::cos(angle), s = std::sin(angle), ic = 1.0f - c;
    rfw_mat4 m = mat4_identity();
    m.m[0] = (c + a.x * a.x * ic) * scale; m.m[1] = (a.y * a.x * ic + a.z * s) * scale; m.m[2] = (a.z * a.x * ic - a.y * s) * scale;
    m.m[4] = (a.x * a.y * ic - a.z * s) * scale; m.m[5] = (c + a.y * a.y * ic) * scale; m.m[6] = (a.z * a.y * ic + a.x * s) * scale;
    m.m[8] = (a.x * a.z * ic + a.y * s) * scale; m.m[9] = (a.y * a.z * ic - a.x * s) * scale; m.m[10] = (c + a.z * a.z * ic) * scale;
    m.m[12] = t[0]; m.m[13] = t[1]; m.m[14] = t[2];
    return m;
}

// ------------------------------------------------------------------ rfw/src/system/mod.rs:19-206
void synchronize_system(Scene& scene, Backend& renderer)
{
    bool changed = false;
    if (scene.skins_changed) { // :36 set_skins comes first
        std::vector<rfw_skin_data> sk;
        for (const Skin& k : scene.skins)
            sk.push_back(rfw_skin_data{k.inverse_bind_matrices.data(), (uint32_t)k.inverse_bind_matrices.size(), k.joint_matrices.data(),
                                       (uint32_t)k.joint_matrices.size()});
        renderer.set_skins(sk, nullptr);
        scene.skins_changed = false;
        changed = true;
    }
    if (!scene.removed_meshes.empty()) {
        renderer.unload_3d_meshes(std::vector<size_t>(scene.removed_meshes.begin(), scene.removed_meshes.end()));
        scene.removed_meshes.clear();
        changed = true;
    }
    for (auto& kv : scene.meshes_3d) { // :63-76 changed meshes
        if (!scene.mesh_changed[kv.first]) continue;
        renderer.set_3d_mesh(kv.first, kv.second.as_data());
        scene.mesh_changed[kv.first] = false;
        changed = true;
    }
    for (auto& kv : scene.instances_3d) { // :82-112 instance lists with any flag set
        if (!scene.instances_changed[kv.first]) continue;
        auto mit = scene.meshes_3d.find(kv.first);
        if (mit == scene.meshes_3d.end()) continue;
        rfw_instances_data_3d d;
        std::memset(&d, 0, sizeof(d));
        d.local_aabb = mit->second.bounds;
        d.matrices = kv.second.matrices.data(); d.num_matrices = (uint32_t)kv.second.matrices.size();
        d.skin_ids = kv.second.skin_ids.data(); d.num_skin_ids = (uint32_t)kv.second.skin_ids.size();
        d.flags = kv.second.flags.data(); d.num_flags = (uint32_t)kv.second.flags.size();
        renderer.set_3d_instances(kv.first, d);
        for (auto& f : kv.second.flags) f = 0;
        scene.instances_changed[kv.first] = false;
        changed = true;
    }
    if (scene.textures_changed) { // :118-136
        std::vector<rfw_texture_data> t;
        for (const Texture& x : scene.textures) t.push_back(x.as_data());
        renderer.set_textures(t, scene.texture_changed_bits.empty() ? nullptr : &scene.texture_changed_bits);
        scene.texture_changed_bits.clear();
        scene.textures_changed = false;
        changed = true;
    }
    if (scene.skybox_changed) {
        renderer.set_skybox(scene.skybox.as_data());
        scene.skybox_changed = false;
        changed = true;
    }
    if (scene.materials_changed) { // :138-147
        renderer.set_materials(scene.device_materials(), scene.material_changed_bits.empty() ? nullptr : &scene.material_changed_bits);
        scene.material_changed_bits.clear();
        scene.materials_changed = false;
        changed = true;
    }
    if (scene.lights_changed) { // :159-190
        renderer.set_point_lights(scene.point_lights, nullptr);
        renderer.set_spot_lights(scene.spot_lights, nullptr);
        renderer.set_area_lights(scene.area_lights, nullptr);
        renderer.set_directional_lights(scene.directional_lights, nullptr);
        scene.lights_changed = false;
        changed = true;
    }
    if (changed) renderer.synchronize(); // :203-205
}

void render_system(const Camera3D& camera, uint32_t width, uint32_t height, Backend& renderer)
{
    const rfw_camera_view_3d view = camera.get_view(width, height);
    renderer.render(mat4_identity(), view, RFW_RENDER_DEFAULT);
}

// ------------------------------------------------------------------ geometry helpers
namespace {
struct Builder {
    MeshDescriptor d;
    void tri(V3 a, V3 b, V3 c, int mat, rfw_vec2 ua = {0, 0}, rfw_vec2 ub = {1, 0}, rfw_vec2 uc = {0, 1})
    {
        const V3 e1 = b - a, e2 = c - a;
        const V3 n = cross(e1, e2);
        if (!(dot(n, n) > 1e-24f)) return; // skip degenerate
        // tangent from uv derivatives (what l3d hands to MeshDescriptor.tangents), w = handedness
        const float du1 = ub.x - ua.x, dv1 = ub.y - ua.y, du2 = uc.x - ua.x, dv2 = uc.y - ua.y;
        float det = du1 * dv2 - du2 * dv1;
        V3 t;
        if (std::fabs(det) > 1e-12f) t = (e1 * dv2 - e2 * dv1) * (1.0f / det);
        else t = e1;
        if (!(dot(t, t) > 1e-24f)) t = e1;
        t = normalize(t);
        const rfw_vec4 tan4{t.x, t.y, t.z, 1.0f};
        const V3 ps[3] = {a, b, c};
        const rfw_vec2 us[3] = {ua, ub, uc};
        for (int i = 0; i < 3; i++) {
            d.vertices.push_back(rfw_vec4{ps[i].x, ps[i].y, ps[i].z, 1.0f});
            d.normals.push_back(rfw_vec3{0, 0, 0});
            d.uvs.push_back(us[i]);
            d.tangents.push_back(tan4);
            d.material_ids.push_back(mat);
        }
    }
    void tri_n(V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, int mat, rfw_vec2 ua, rfw_vec2 ub, rfw_vec2 uc)
    {
        const size_t before = d.vertices.size();
        tri(a, b, c, mat, ua, ub, uc);
        if (d.vertices.size() == before) return;
        d.normals[before] = pod(na); d.normals[before + 1] = pod(nb); d.normals[before + 2] = pod(nc);
    }
    void quad(V3 a, V3 b, V3 c, V3 dd, int mat)
    {
        tri(a, b, c, mat, {0, 0}, {1, 0}, {1, 1});
        tri(a, c, dd, mat, {0, 0}, {1, 1}, {0, 1});
    }
    // axis-aligned box with outward-facing normals
    void box(V3 lo, V3 hi, int mat, const rfw_mat4* xf = nullptr)
    {
        V3 c[8];
        for (int i = 0; i < 8; i++) {
            c[i] = v3((i & 1) ? hi.x : lo.x, (i & 2) ? hi.y : lo.y, (i & 4) ? hi.z : lo.z);
            if (xf) c[i] = mul_point(*xf, c[i]);
        }
        quad(c[0], c[2], c[3], c[1], mat); // -z
        quad(c[4], c[5], c[7], c[6], mat); // +z
        quad(c[0], c[4], c[6], c[2], mat); // -x
        quad(c[1], c[3], c[7], c[5], mat); // +x
        quad(c[0], c[1], c[5], c[4], mat); // -y
        quad(c[2], c[6], c[7], c[3], mat); // +y
    }
    // parametric surface f(u,v) on [0,1]^2 tessellated nu x nv, smooth normals by finite differences when smooth=true
    template <typename F> void surface(F f, int nu, int nv, int mat, bool flip = false)
    {
        nu = std::max(nu, 1); nv = std::max(nv, 1);
        for (int j = 0; j < nv; j++) {
            for (int i = 0; i < nu; i++) {
                const float u0 = (float)i / nu, u1 = (float)(i + 1) / nu, v0 = (float)j / nv, v1 = (float)(j + 1) / nv;
                const V3 a = f(u0, v0), b = f(u1, v0), c = f(u1, v1), dd = f(u0, v1);
                if (!flip) {
                    tri(a, b, c, mat, {u0, v0}, {u1, v0}, {u1, v1});
                    tri(a, c, dd, mat, {u0, v0}, {u1, v1}, {u0, v1});
                } else {
                    tri(a, c, b, mat, {u0, v0}, {u1, v1}, {u1, v0});
                    tri(a, dd, c, mat, {u0, v0}, {u0, v1}, {u1, v1});
                }
            }
        }
    }
};

float hash2(int x, int y, uint32_t seed)
{
    uint32_t h = (uint32_t)x * 374761393u + (uint32_t)y * 668265263u + seed * 2246822519u;
    h = (h ^ (h >> 13)) * 1274126177u;
    h ^= h >> 16;
    return (float)(h & 0xffffff) * (1.0f / 16777216.0f);
}
float vnoise(float x, float y, uint32_t seed)
{
    const float fx = std::floor(x), fy = std::floor(y);
    const int ix = (int)fx, iy = (int)fy;
    float tx = x - fx, ty = y - fy;
    tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty);
    const float a = hash2(ix, iy, seed), b = hash2(ix + 1, iy, seed), c = hash2(ix, iy + 1, seed), d = hash2(ix + 1, iy + 1, seed);
    return (a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty;
}
} // namespace

// objects_3d/quad.rs:52-75 Quad3D::generate_render_data: two triangles spanned by a tangent (normal x helper axis, half the width) and a
// bitangent (half the height) around `position`; uvs are all zero there ("TODO: uvs", :28)
MeshDescriptor make_quad(const float n_in[3], const float p_in[3], float width, float height, uint32_t mat_id)
{
    const V3 normal = normalize(v3(n_in[0], n_in[1], n_in[2])), pos = v3(p_in[0], p_in[1], p_in[2]);
    const V3 tmp = normal.x > 0.9f ? v3(0, 1, 0) : v3(1, 0, 0);
    const V3 tangent = normalize(cross(normal, tmp)) * (0.5f * width);
    const V3 bi_tangent = cross(normalize(tangent), normal) * (0.5f * height);
    const V3 v[6] = {pos - bi_tangent - tangent, pos + bi_tangent - tangent, pos - bi_tangent + tangent,
                     pos + bi_tangent - tangent, pos + bi_tangent + tangent, pos - bi_tangent + tangent};
    MeshDescriptor d;
    d.name = "quad";
    for (const V3& q : v) {
        d.vertices.push_back(rfw_vec4{q.x, q.y, q.z, 1.0f});
        d.normals.push_back(pod(normal));
        d.uvs.push_back(rfw_vec2{0, 0});
        d.material_ids.push_back((int32_t)mat_id);
    }
    return d;
}

MeshDescriptor make_icosphere(int quality, uint32_t mat_id)
{
    const float PI = 3.14159265358979323846f;
    std::vector<V3> vertices;
    std::vector<std::array<uint32_t, 3>> faces;
    const float s = std::sqrt((5.0f - std::sqrt(5.0f)) / 10.0f), t = std::sqrt((5.0f + std::sqrt(5.0f)) / 10.0f);
    auto add_vertex = [&](V3 v) { vertices.push_back(normalize(v)); return (uint32_t)vertices.size() - 1; };
    add_vertex(v3(-s, t, 0)); add_vertex(v3(s, t, 0)); add_vertex(v3(-s, -t, 0)); add_vertex(v3(s, -t, 0));
    add_vertex(v3(0, -s, t)); add_vertex(v3(0, s, t)); add_vertex(v3(0, -s, -t)); add_vertex(v3(0, s, -t));
    add_vertex(v3(t, 0, -s)); add_vertex(v3(t, 0, s)); add_vertex(v3(-t, 0, -s)); add_vertex(v3(-t, 0, s));
    const uint32_t f0[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                {3, 9, 4},  {3, 4, 2}, {3, 2, 6}, {3, 6, 8},  {3, 8, 9},   {4, 9, 5}, {2, 4, 11}, {6, 2, 10},  {8, 6, 7},  {9, 8, 1}};
    for (auto& f : f0) faces.push_back({f[0], f[1], f[2]});
    std::unordered_map<uint64_t, uint32_t> mid;
    auto middle = [&](uint32_t a, uint32_t b) {
        const uint64_t key = ((uint64_t)std::min(a, b) << 32) + std::max(a, b);
        auto it = mid.find(key);
        if (it != mid.end()) return it->second;
        const uint32_t idx = add_vertex((vertices[a] + vertices[b]) * 0.5f);
        mid[key] = idx;
        return idx;
    };
    for (int q = 0; q < quality; q++) {
        std::vector<std::array<uint32_t, 3>> nf;
        nf.reserve(faces.size() * 4);
        for (auto& f : faces) {
            const uint32_t a = middle(f[0], f[1]), b = middle(f[1], f[2]), c = middle(f[2], f[0]);
            nf.push_back({f[0], a, c}); nf.push_back({f[1], b, a}); nf.push_back({f[2], c, b}); nf.push_back({a, b, c});
        }
        faces.swap(nf);
    }
    MeshDescriptor d;
    d.name = "sphere";
    for (auto& f : faces) {
        for (int k = 0; k < 3; k++) {
            const V3 v = vertices[f[k]];
            d.vertices.push_back(rfw_vec4{v.x, v.y, v.z, 1.0f});
            d.normals.push_back(pod(v));
            d.uvs.push_back(rfw_vec2{std::atan2(v.x, v.z) / (2.0f * PI) + 0.5f, v.y * 0.5f + 0.5f});
            V3 tg = cross(v3(0, 1, 0), v);
            if (!(dot(tg, tg) > 1e-12f)) tg = v3(1, 0, 0);
            tg = normalize(tg);
            d.tangents.push_back(rfw_vec4{tg.x, tg.y, tg.z, 1.0f});
            d.material_ids.push_back((int32_t)mat_id);
        }
    }
    return d;
}

// C1: Cornell box in [-1,1]^3, 36 triangles; materials from assets/models/cbox.mtl:4-40
void build_cornell_box(Scene& scene, Camera3D& cam)
{
    Material khaki; khaki.color[0] = 0.8f; khaki.color[1] = 0.659f; khaki.color[2] = 0.44f; khaki.roughness = 1.0f; khaki.specular_f = 0.0f;
    Material red = khaki; red.color[0] = 0.445f; red.color[1] = 0.0f; red.color[2] = 0.0f;
    Material green = khaki; green.color[0] = 0.0f; green.color[1] = 0.32f; green.color[2] = 0.0f;
    Material light = khaki; light.color[0] = 10.0f; light.color[1] = 10.0f; light.color[2] = 10.0f;
    const int m_khaki = (int)scene.add_material(khaki), m_red = (int)scene.add_material(red), m_green = (int)scene.add_material(green),
              m_light = (int)scene.add_material(light);
    Builder b;
    b.d.name = "cornell";
    // inward-facing walls (camera at z = -3.4 looks along +z; the -z side is open)
    b.quad(v3(-1, -1, -1), v3(-1, -1, 1), v3(1, -1, 1), v3(1, -1, -1), m_khaki); // floor, normal +y
    b.quad(v3(-1, 1, -1), v3(1, 1, -1), v3(1, 1, 1), v3(-1, 1, 1), m_khaki);     // ceiling, normal -y
    b.quad(v3(-1, -1, 1), v3(-1, 1, 1), v3(1, 1, 1), v3(1, -1, 1), m_khaki);     // back, normal -z
    b.quad(v3(-1, -1, -1), v3(-1, 1, -1), v3(-1, 1, 1), v3(-1, -1, 1), m_red);   // left, normal +x
    b.quad(v3(1, -1, -1), v3(1, -1, 1), v3(1, 1, 1), v3(1, 1, -1), m_green);     // right, normal -x
    b.quad(v3(-0.25f, 0.995f, -0.25f), v3(0.25f, 0.995f, -0.25f), v3(0.25f, 0.995f, 0.25f), v3(-0.25f, 0.995f, 0.25f), m_light); // light, normal -y
    const float t0[3] = {0.35f, -0.7f, -0.2f}, ax[3] = {0, 1, 0};
    rfw_mat4 m_short = mat4_from_trs(t0, ax, -0.3f, 1.0f);
    b.box(v3(-0.3f, -0.3f, -0.3f), v3(0.3f, 0.3f, 0.3f), m_khaki, &m_short);
    const float t1[3] = {-0.35f, -0.4f, 0.35f};
    rfw_mat4 m_tall = mat4_from_trs(t1, ax, 0.3f, 1.0f);
    b.box(v3(-0.3f, -0.6f, -0.3f), v3(0.3f, 0.6f, 0.3f), m_khaki, &m_tall);
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(b.d));
    scene.add_instance(mesh, mat4_identity());
    scene.update_lights();
    cam = Camera3D();
    cam.pos[0] = 0; cam.pos[1] = 0; cam.pos[2] = -3.4f;
    cam.direction[0] = 0; cam.direction[1] = 0; cam.direction[2] = 1;
    cam.fov = 40.0f;
    cam.aperture = 0.0f;
}

namespace {
// atrium ("Sponza-class") pieces; `s` scales the tessellation of every part, triangle count ~ s^2
void atrium_geometry(Builder& b, float s, uint32_t seed, const std::vector<int>& mats, int m_light)
{
    auto T = [&](float base) { return std::max(1, (int)std::lround(base * s)); };
    const float X = 15.0f, Z = 6.0f, H = 12.0f;
    const float PI = 3.14159265358979323846f;
    int mi = 0;
    auto next_mat = [&]() { return mats[(mi++) % mats.size()]; };
    // floor with shallow relief (tiles)
    b.surface([&](float u, float v) { const float x = -X + 2 * X * u, z = -Z + 2 * Z * v; return v3(x, 0.03f * vnoise(x * 2.0f, z * 2.0f, seed), z); },
              T(60), T(24), next_mat(), true);
    // long walls (z = +-Z) with brick relief, short walls (x = +-X)
    for (int side = 0; side < 2; side++) {
        const float zs = side ? Z : -Z;
        const int m = next_mat();
        b.surface([&](float u, float v) {
            const float x = -X + 2 * X * u, y = H * v;
            const float r = 0.05f * vnoise(x * 3.0f, y * 3.0f, seed + 7 + side);
            return v3(x, y, zs + (side ? -r : r));
        }, T(60), T(24), m, side == 0);
        const float xs = side ? X : -X;
        const int m2 = next_mat();
        b.surface([&](float u, float v) {
            const float z = -Z + 2 * Z * u, y = H * v;
            const float r = 0.05f * vnoise(z * 3.0f, y * 3.0f, seed + 11 + side);
            return v3(xs + (side ? -r : r), y, z);
        }, T(24), T(24), m2, side == 1);
    }
    // ceiling ring around the open court (court: |x| < 10, |z| < 3)
    {
        const int m = next_mat();
        b.surface([&](float u, float v) { return v3(-X + 2 * X * u, H, -Z + 3.0f * v); }, T(60), T(6), m, false);
        b.surface([&](float u, float v) { return v3(-X + 2 * X * u, H, 3.0f + 3.0f * v); }, T(60), T(6), m, false);
        b.surface([&](float u, float v) { return v3(-X + 5.0f * u, H, -3.0f + 6.0f * v); }, T(10), T(12), m, false);
        b.surface([&](float u, float v) { return v3(10.0f + 5.0f * u, H, -3.0f + 6.0f * v); }, T(10), T(12), m, false);
    }
    // gallery slabs at y = 5 (top and underside)
    {
        const int m = next_mat();
        for (int side = 0; side < 2; side++) {
            const float z0 = side ? 3.0f : -Z, z1 = side ? Z : -3.0f;
            b.surface([&](float u, float v) { return v3(-X + 2 * X * u, 5.0f, z0 + (z1 - z0) * v); }, T(60), T(6), m, true);
            b.surface([&](float u, float v) { return v3(-X + 2 * X * u, 4.7f, z0 + (z1 - z0) * v); }, T(60), T(6), m, false);
            const float zi = side ? 3.0f : -3.0f;
            b.surface([&](float u, float v) { return v3(-X + 2 * X * u, 4.7f + 0.3f * v, zi); }, T(60), T(1), m, side == 1);
        }
    }
    // columns: two rows x two storeys, fluted shafts
    {
        const int m_col = next_mat(), m_cap = next_mat();
        for (int row = 0; row < 2; row++) {
            const float zc = row ? 3.0f : -3.0f;
            for (int k = 0; k < 9; k++) {
                const float xc = -12.0f + 3.0f * k;
                for (int storey = 0; storey < 2; storey++) {
                    const float y0 = storey ? 5.0f : 0.0f, y1 = storey ? 9.5f : 4.7f;
                    b.surface([&](float u, float v) {
                        const float a = 2 * PI * u;
                        const float flute = 1.0f + 0.04f * std::cos(a * 16.0f);
                        const float taper = 1.0f - 0.12f * v;
                        const float r = 0.33f * flute * taper;
                        return v3(xc + r * std::cos(a), y0 + (y1 - y0) * v, zc + r * std::sin(a));
                    }, T(40), T(12), m_col, true);
                    // capital + base as boxes
                    b.box(v3(xc - 0.45f, y1 - 0.25f, zc - 0.45f), v3(xc + 0.45f, y1, zc + 0.45f), m_cap);
                    b.box(v3(xc - 0.45f, y0, zc - 0.45f), v3(xc + 0.45f, y0 + 0.2f, zc + 0.45f), m_cap);
                }
            }
        }
    }
    // arches between ground-storey columns (half tori)
    {
        const int m = next_mat();
        for (int row = 0; row < 2; row++) {
            const float zc = row ? 3.0f : -3.0f;
            for (int k = 0; k < 8; k++) {
                const float xc = -10.5f + 3.0f * k;
                b.surface([&](float u, float v) {
                    const float a = PI * u, bb = 2 * PI * v;
                    const float R = 1.17f, r = 0.18f;
                    return v3(xc + (R + r * std::cos(bb)) * std::cos(a), 3.4f + (R + r * std::cos(bb)) * std::sin(a), zc + r * std::sin(bb));
                }, T(24), T(10), m, false);
            }
        }
    }
    // drapes hanging from the upper gallery: sheets with folds (the dense, thin geometry Sponza is known for)
    {
        for (int k = 0; k < 8; k++) {
            const int m = next_mat();
            const float xc = -10.5f + 3.0f * k;
            const float zc = (k & 1) ? 2.6f : -2.6f;
            const float phase = hash2(k, 3, seed) * 6.28f;
            b.surface([&](float u, float v) {
                const float x = xc - 1.2f + 2.4f * u;
                const float y = 9.3f - 3.8f * v;
                const float fold = 0.12f * std::sin(u * 22.0f + phase) * (0.3f + v) + 0.05f * std::sin(v * 9.0f + u * 5.0f);
                return v3(x, y, zc + fold);
            }, T(56), T(64), m, (k & 1) != 0);
        }
    }
    // balustrade: rows of small balusters along the gallery edge
    {
        const int m = next_mat();
        for (int row = 0; row < 2; row++) {
            const float zc = row ? 3.15f : -3.15f;
            for (int k = 0; k < 96; k++) {
                const float xc = -14.2f + 0.3f * k;
                b.surface([&](float u, float v) {
                    const float a = 2 * PI * u;
                    const float r = 0.04f + 0.025f * std::sin(v * PI * 3.0f) * std::sin(v * PI);
                    return v3(xc + r * std::cos(a), 5.0f + 0.9f * v, zc + r * std::sin(a));
                }, T(8), T(6), m, true);
            }
            b.box(v3(-14.4f, 5.9f, zc - 0.08f), v3(14.4f, 6.0f, zc + 0.08f), m);
        }
    }
    // clutter: vases (surfaces of revolution) and crates on the court floor
    {
        Rng rng(seed + 99);
        for (int k = 0; k < 14; k++) {
            const int m = next_mat();
            const float xc = rng.range(-9.0f, 9.0f), zc = rng.range(-2.3f, 2.3f), sc = rng.range(0.5f, 1.1f);
            if (k & 1) {
                b.surface([&](float u, float v) {
                    const float a = 2 * PI * u;
                    const float r = sc * (0.18f + 0.22f * std::sin(v * PI) * (1.0f - 0.5f * v));
                    return v3(xc + r * std::cos(a), 0.03f + sc * 1.1f * v, zc + r * std::sin(a));
                }, T(24), T(16), m, true);
            } else {
                const float t[3] = {xc, 0.03f + 0.35f * sc, zc}, ax[3] = {0, 1, 0};
                rfw_mat4 xf = mat4_from_trs(t, ax, rng.range(0, 3.0f), sc);
                b.box(v3(-0.35f, -0.35f, -0.35f), v3(0.35f, 0.35f, 0.35f), m, &xf);
            }
        }
    }
    // two emissive panels under the ceiling ring (area lights)
    b.quad(v3(-13.0f, H - 0.05f, -5.0f), v3(-11.0f, H - 0.05f, -5.0f), v3(-11.0f, H - 0.05f, -4.0f), v3(-13.0f, H - 0.05f, -4.0f), m_light);
    b.quad(v3(11.0f, H - 0.05f, 4.0f), v3(13.0f, H - 0.05f, 4.0f), v3(13.0f, H - 0.05f, 5.0f), v3(11.0f, H - 0.05f, 5.0f), m_light);
}
} // namespace

// C2 ("Sponza-class", target 262 267) and C4 (target ~1 048 576: atrium + 64 displaced icospheres)
// One of C4's 64 displaced icospheres, in place (k: 16 per row along x, two rows in z, two storeys)
MeshDescriptor make_displaced_sphere(int k, uint32_t seed, int quality, uint32_t mat_id)
{
    MeshDescriptor out;
    out.name = "displaced-sphere";
    const MeshDescriptor base = make_icosphere(quality, mat_id);
    const float cx = -13.0f + 26.0f * ((k % 16) + 0.5f) / 16.0f, cz = ((k / 16) % 2 ? 4.6f : -4.6f), cy = (k / 32) ? 5.6f : 0.65f;
    const float rad = 0.55f;
    for (size_t i = 0; i < base.vertices.size(); i++) {
        const V3 n = v3(base.vertices[i].x, base.vertices[i].y, base.vertices[i].z);
        const float disp = 1.0f + 0.18f * vnoise(n.x * 4.0f + 10.0f, n.y * 4.0f + n.z * 3.0f + 10.0f, seed);
        const V3 p = v3(cx, cy, cz) + n * (rad * disp);
        out.vertices.push_back(rfw_vec4{p.x, p.y, p.z, 1.0f});
        out.normals.push_back(rfw_vec3{0, 0, 0}); // regenerate: displacement changed the surface
        out.uvs.push_back(base.uvs[i]);
        out.tangents.push_back(base.tangents[i]);
        out.material_ids.push_back((int32_t)mat_id);
    }
    return out;
}

void build_atrium(Scene& scene, Camera3D& cam, uint32_t target_triangles, uint32_t seed, int sphere_meshes)
{
    const bool separate_spheres = sphere_meshes == 1, one_mesh = sphere_meshes == 2;
    std::vector<int> mats;
    Rng rng(seed);
    for (int i = 0; i < 23; i++) {
        Material m;
        m.color[0] = rng.range(0.25f, 0.9f); m.color[1] = rng.range(0.25f, 0.9f); m.color[2] = rng.range(0.25f, 0.9f);
        const float rough[] = {0.1f, 0.25f, 0.4f, 0.55f, 0.7f, 0.85f, 1.0f};
        m.roughness = rough[i % 7];
        m.metallic = (i % 5 == 4) ? 1.0f : 0.0f;
        m.specular_f = 0.5f;
        m.clearcoat = (i % 6 == 3) ? 0.5f : 0.0f;
        mats.push_back((int)scene.add_material(m));
    }
    Material light; light.color[0] = 24.0f; light.color[1] = 22.0f; light.color[2] = 18.0f;
    const int m_light = (int)scene.add_material(light);
    Material sphere_mat; sphere_mat.color[0] = 0.8f; sphere_mat.color[1] = 0.75f; sphere_mat.color[2] = 0.6f; sphere_mat.roughness = 0.3f;
    const int m_sphere = (int)scene.add_material(sphere_mat);

    const bool with_spheres = target_triangles > 600000u;
    const uint32_t sphere_tris = with_spheres ? 64u * 5120u : 0u;
    const uint32_t atrium_target = target_triangles - sphere_tris;
    // triangle count ~ s^2: secant iterations on s
    float s = 1.0f;
    size_t count = 0;
    for (int it = 0; it < 6; it++) {
        Builder probe;
        atrium_geometry(probe, s, seed, mats, m_light);
        count = probe.d.vertices.size() / 3;
        const float ratio = (float)atrium_target / (float)count;
        if (std::fabs(ratio - 1.0f) < 0.002f) break;
        s *= std::sqrt(ratio);
    }
    Builder b;
    b.d.name = "atrium";
    atrium_geometry(b, s, seed, mats, m_light);
    if (with_spheres && one_mesh) { // (measurement aid) the spheres baked into the atrium's own mesh: what a single-level BVH would traverse
        for (int k = 0; k < 64; k++) {
            const MeshDescriptor sp = make_displaced_sphere(k, seed + 1000 + k, 4, (uint32_t)mats[(k * 7) % mats.size()]);
            b.d.vertices.insert(b.d.vertices.end(), sp.vertices.begin(), sp.vertices.end());
            b.d.normals.insert(b.d.normals.end(), sp.normals.begin(), sp.normals.end());
            b.d.uvs.insert(b.d.uvs.end(), sp.uvs.begin(), sp.uvs.end());
            b.d.tangents.insert(b.d.tangents.end(), sp.tangents.begin(), sp.tangents.end());
            b.d.material_ids.insert(b.d.material_ids.end(), sp.material_ids.begin(), sp.material_ids.end());
        }
    }
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(b.d));
    scene.add_instance(mesh, mat4_identity());

    if (with_spheres && one_mesh) {
    } else if (with_spheres && separate_spheres) {
        // SURVEY §8d C4 literally: 64 displaced icospheres as 64 meshes, one instance each
        for (int k = 0; k < 64; k++) {
            const uint32_t m2 = scene.add_mesh(Mesh3D::from(make_displaced_sphere(k, seed + 1000 + k, 4, (uint32_t)mats[(k * 7) % mats.size()])));
            scene.add_instance(m2, mat4_identity());
        }
    } else if (with_spheres) {
        // 64 displaced icospheres (Quality::VeryHigh = 5120 triangles each), baked into one static mesh
        MeshDescriptor all;
        all.name = "displaced-spheres";
        const MeshDescriptor base = make_icosphere(4, (uint32_t)m_sphere);
        Rng r2(seed + 5);
        for (int k = 0; k < 64; k++) {
            const float cx = -13.0f + 26.0f * ((k % 16) + 0.5f) / 16.0f, cz = ((k / 16) % 2 ? 4.6f : -4.6f), cy = (k / 32) ? 5.6f : 0.65f;
            const float rad = 0.55f;
            const uint32_t sd = seed + 1000 + k;
            for (size_t i = 0; i < base.vertices.size(); i++) {
                const V3 n = v3(base.vertices[i].x, base.vertices[i].y, base.vertices[i].z);
                const float disp = 1.0f + 0.18f * vnoise(n.x * 4.0f + 10.0f, n.y * 4.0f + n.z * 3.0f + 10.0f, sd);
                const V3 p = v3(cx, cy, cz) + n * (rad * disp);
                all.vertices.push_back(rfw_vec4{p.x, p.y, p.z, 1.0f});
                all.normals.push_back(rfw_vec3{0, 0, 0}); // regenerate: displacement changed the surface
                all.uvs.push_back(base.uvs[i]);
                all.tangents.push_back(base.tangents[i]);
                all.material_ids.push_back(mats[(k * 7) % mats.size()]);
            }
        }
        (void)r2;
        const uint32_t m2 = scene.add_mesh(Mesh3D::from(all));
        scene.add_instance(m2, mat4_identity());
    }
    // sun through the open court
    rfw_directional_light sun;
    std::memset(&sun, 0, sizeof(sun));
    const V3 dir = normalize(v3(0.25f, -1.0f, 0.18f));
    sun.direction = pod(dir);
    sun.radiance = rfw_vec3{6.0f, 5.6f, 5.0f};
    sun.energy = length(v3(6.0f, 5.6f, 5.0f));
    scene.directional_lights.push_back(sun);
    scene.update_lights();

    cam = Camera3D();
    cam.pos[0] = -13.0f; cam.pos[1] = 2.2f; cam.pos[2] = 0.4f;
    const V3 d = normalize(v3(1.0f, 0.12f, -0.03f));
    cam.direction[0] = d.x; cam.direction[1] = d.y; cam.direction[2] = d.z;
    cam.fov = 60.0f;
    cam.aperture = 0.0f;
    cam.aspect_ratio = 1920.0f / 1080.0f;
}

// C3: nx*nz instances of the 320-triangle icosphere on a grid (examples/animated/src/main.rs:35-118)
void add_sphere_grid(Scene& scene, uint32_t nx, uint32_t nz, float spacing)
{
    Material m; m.color[0] = 0.9f; m.color[1] = 0.3f; m.color[2] = 0.25f; m.roughness = 0.4f;
    const uint32_t mat = scene.add_material(m);
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(make_icosphere(2, mat)));
    for (uint32_t x = 0; x < nx; x++)
        for (uint32_t z = 0; z < nz; z++) scene.add_instance(mesh, mat4_identity());
    animate_sphere_grid(scene, mesh, nx, nz, spacing, 0.0f);
}
// examples/animated/src/main.rs:197-219: y = 0.3 + ((sin(x + t) + sin(z + t)) * 0.5 + 1)
void animate_sphere_grid(Scene& scene, uint32_t mesh, uint32_t nx, uint32_t nz, float spacing, float time)
{
    const float r = 0.4f * spacing;
    for (uint32_t x = 0; x < nx; x++) {
        for (uint32_t z = 0; z < nz; z++) {
            const float px = ((float)x - 0.5f * (float)(nx - 1)) * spacing;
            const float pz = ((float)z - 0.5f * (float)(nz - 1)) * spacing;
            const float height = (std::sin((float)x + time) + std::sin((float)z + time)) * 0.5f + 1.0f;
            scene.set_matrix(mesh, (size_t)x * nz + z, mat4_from_scale_translation(r, px, 0.3f + height * spacing, pz));
        }
    }
}

// random triangle soups + transformed instances: the BVH-vs-brute-force and CPU-vs-GPU equivalence inputs
void build_soup(Scene& scene, Camera3D& cam, uint32_t triangles, uint32_t instances, uint32_t seed)
{
    Rng rng(seed);
    std::vector<int> mats;
    for (int i = 0; i < 6; i++) {
        Material m;
        m.color[0] = rng.range(0.2f, 0.95f); m.color[1] = rng.range(0.2f, 0.95f); m.color[2] = rng.range(0.2f, 0.95f);
        m.roughness = rng.range(0.05f, 1.0f);
        m.metallic = (i == 2) ? 1.0f : 0.0f;
        m.subsurface = (i == 3) ? 0.4f : 0.0f;
        m.clearcoat = (i == 4) ? 1.0f : 0.0f;
        m.transmission = (i == 5) ? 0.6f : 0.0f;
        m.eta = (i == 5) ? 0.66f : 1.0f;
        if (i == 5) { m.absorption[0] = 0.2f; m.absorption[1] = 0.05f; m.absorption[2] = 0.4f; }
        mats.push_back((int)scene.add_material(m));
    }
    Material light; light.color[0] = 12.0f; light.color[1] = 11.0f; light.color[2] = 9.0f;
    const int m_light = (int)scene.add_material(light);
    Builder b;
    b.d.name = "soup";
    for (uint32_t i = 0; i < triangles; i++) {
        const V3 c = v3(rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1));
        const float sz = rng.range(0.02f, 0.35f);
        const V3 p0 = c + v3(rng.range(-sz, sz), rng.range(-sz, sz), rng.range(-sz, sz));
        const V3 p1 = c + v3(rng.range(-sz, sz), rng.range(-sz, sz), rng.range(-sz, sz));
        const V3 p2 = c + v3(rng.range(-sz, sz), rng.range(-sz, sz), rng.range(-sz, sz));
        b.tri(p0, p1, p2, (i % 37 == 0) ? m_light : mats[i % mats.size()]);
    }
    // a few exactly shared edges / coplanar duplicates to exercise the tie rule
    b.quad(v3(-1.5f, -1.2f, -1.5f), v3(-1.5f, -1.2f, 1.5f), v3(1.5f, -1.2f, 1.5f), v3(1.5f, -1.2f, -1.5f), mats[0]);
    b.quad(v3(-0.5f, -1.2f, -0.5f), v3(-0.5f, -1.2f, 0.5f), v3(0.5f, -1.2f, 0.5f), v3(0.5f, -1.2f, -0.5f), mats[1]);
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(b.d));
    scene.add_instance(mesh, mat4_identity());
    for (uint32_t k = 1; k < instances; k++) {
        const float t[3] = {rng.range(-3, 3), rng.range(-1, 2), rng.range(-3, 3)};
        const float ax[3] = {rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1) + 1.5f};
        scene.add_instance(mesh, mat4_from_trs(t, ax, rng.range(0, 6.28f), rng.range(0.3f, 1.4f)));
    }
    rfw_point_light pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.position = rfw_vec3{0.5f, 3.5f, -2.0f};
    pl.radiance = rfw_vec3{20, 20, 20};
    pl.energy = length(v3(20, 20, 20));
    scene.point_lights.push_back(pl);
    rfw_spot_light sl;
    std::memset(&sl, 0, sizeof(sl));
    sl.position = rfw_vec3{-3.0f, 4.0f, -3.0f};
    const V3 sd = normalize(v3(0.6f, -0.8f, 0.6f));
    sl.direction = pod(sd);
    sl.radiance = rfw_vec3{30, 28, 25};
    sl.energy = length(v3(30, 28, 25));
    sl.cos_inner = 0.92f; sl.cos_outer = 0.8f;
    scene.spot_lights.push_back(sl);
    scene.update_lights();
    cam = Camera3D();
    cam.pos[0] = 0.3f; cam.pos[1] = 0.8f; cam.pos[2] = -6.5f;
    const V3 d = normalize(v3(-0.03f, -0.08f, 1.0f));
    cam.direction[0] = d.x; cam.direction[1] = d.y; cam.direction[2] = d.z;
    cam.fov = 55.0f;
    cam.aperture = 0.02f; // thin lens: exercises the 9-blade aperture code
    cam.focal_distance = 1.0f;
}

namespace {
Texture make_texture(uint32_t w, uint32_t h, uint32_t seed, int kind)
{
    Texture t;
    t.width = w; t.height = h; t.format = RFW_FORMAT_BGRA8;
    t.bytes.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* px = &t.bytes[((size_t)y * w + x) * 4];
            if (kind == 0) { // checker with noise (albedo), BGRA
                const bool c = (((x * 8) / w) + ((y * 8) / h)) & 1;
                const float n = vnoise(x * 0.37f, y * 0.37f, seed);
                px[0] = (uint8_t)(c ? 60 + 40 * n : 200 + 50 * n);
                px[1] = (uint8_t)(c ? 90 + 60 * n : 190 + 40 * n);
                px[2] = (uint8_t)(c ? 200 + 50 * n : 120 + 60 * n);
                px[3] = 255;
            } else if (kind == 1) { // normal map: bumps, tangent-space (r = x, g = y, b = z) stored BGRA
                const float fx = std::sin(x * 6.2831853f * 4.0f / w) * 0.35f, fy = std::cos(y * 6.2831853f * 3.0f / h) * 0.35f;
                const float fz = std::sqrt(std::max(0.0f, 1.0f - fx * fx - fy * fy));
                px[2] = (uint8_t)((fx * 0.5f + 0.5f) * 255.0f);
                px[1] = (uint8_t)((fy * 0.5f + 0.5f) * 255.0f);
                px[0] = (uint8_t)((fz * 0.5f + 0.5f) * 255.0f);
                px[3] = 255;
            } else { // sky: vertical gradient + a bright band, RGBA order to exercise the second format
                const float v = (float)y / (float)h;
                t.format = RFW_FORMAT_RGBA8;
                px[0] = (uint8_t)(60 + 120 * v);
                px[1] = (uint8_t)(110 + 100 * v);
                px[2] = (uint8_t)(235 - 60 * v);
                px[3] = 255;
                if (((x * 16) / w) % 5 == 0 && v < 0.45f) { px[0] = 250; px[1] = 240; px[2] = 200; }
            }
        }
    t.generate_mipmaps(5);
    return t;
}
} // namespace

void build_gallery(Scene& scene, Camera3D& cam, uint32_t seed)
{
    scene.textures.push_back(make_texture(64, 64, seed, 0));       // 0 albedo checker
    scene.textures.push_back(make_texture(32, 64, seed + 1, 1));   // 1 normal map (non-square)
    scene.textures.push_back(make_texture(16, 16, seed + 2, 0));   // 2 small albedo (few mips)
    scene.skybox = make_texture(128, 64, seed + 3, 2);
    scene.textures_changed = true;
    scene.skybox_changed = true;
    Material plain; plain.color[0] = 0.75f; plain.color[1] = 0.7f; plain.color[2] = 0.6f; plain.roughness = 0.9f;
    Material tex = plain; tex.color[0] = tex.color[1] = tex.color[2] = 1.0f; tex.diffuse_tex = 0; tex.roughness = 0.6f;
    Material bump = plain; bump.diffuse_tex = 2; bump.normal_tex = 1; bump.roughness = 0.35f; bump.metallic = 0.0f; bump.clearcoat = 0.5f;
    Material emap = plain; emap.color[0] = 4.0f; emap.color[1] = 3.0f; emap.color[2] = 2.0f; emap.emissive_tex = 0; // emissive map: shaded, not a light (shade.comp:128)
    Material light = plain; light.color[0] = 14.0f; light.color[1] = 13.0f; light.color[2] = 11.0f;
    const int m_plain = (int)scene.add_material(plain), m_tex = (int)scene.add_material(tex), m_bump = (int)scene.add_material(bump),
              m_emap = (int)scene.add_material(emap), m_light = (int)scene.add_material(light);
    Builder b;
    b.d.name = "gallery";
    b.surface([&](float u, float v) { return v3(-3 + 6 * u, 0.0f, -3 + 6 * v); }, 6, 6, m_tex, true);                  // textured floor
    b.surface([&](float u, float v) { return v3(-3 + 6 * u, 3 * v, 3.0f); }, 5, 3, m_bump, true);                      // bump-mapped back wall
    b.surface([&](float u, float v) { return v3(-3.0f, 3 * v, -3 + 6 * u); }, 4, 3, m_tex, false);                     // left wall
    b.quad(v3(3, 0.3f, -1.5f), v3(3, 0.3f, 1.5f), v3(3, 2.4f, 1.5f), v3(3, 2.4f, -1.5f), m_emap);                       // emissive-mapped panel
    const float t0[3] = {0.6f, 0.5f, 0.4f}, ax[3] = {0.2f, 1, 0.1f};
    rfw_mat4 xf = mat4_from_trs(t0, ax, 0.6f, 1.0f);
    b.box(v3(-0.5f, -0.5f, -0.5f), v3(0.5f, 0.5f, 0.5f), m_bump, &xf);
    b.quad(v3(-1, 2.95f, -1), v3(1, 2.95f, -1), v3(1, 2.95f, 1), v3(-1, 2.95f, 1), m_light);                            // lamp, faces down
    b.quad(v3(-1.6f, 0.0f, -2.4f), v3(-0.4f, 0.0f, -2.4f), v3(-0.4f, 1.1f, -2.4f), v3(-1.6f, 1.1f, -2.4f), m_plain);
    const uint32_t mesh = scene.add_mesh(Mesh3D::from(b.d));
    scene.add_instance(mesh, mat4_identity());
    const float t1[3] = {-1.6f, 0.0f, 0.9f}, ay[3] = {0, 1, 0};
    scene.add_instance(scene.add_mesh(Mesh3D::from(make_icosphere(2, (uint32_t)m_bump))), mat4_from_trs(t1, ay, 0.4f, 0.6f));
    scene.update_lights();
    cam = Camera3D();
    cam.pos[0] = 0.4f; cam.pos[1] = 1.5f; cam.pos[2] = -5.5f;
    const V3 d = normalize(v3(-0.05f, -0.05f, 1.0f));
    cam.direction[0] = d.x; cam.direction[1] = d.y; cam.direction[2] = d.z;
    cam.fov = 60.0f;
    cam.aperture = 0.0f;
}

void pose_skins(Scene& scene, float time)
{
    // joint k of skin s: rotate about z by an angle growing along the chain, pivoting at the joint's rest position (0, k * 0.5, 0)
    for (size_t si = 0; si < scene.skins.size(); si++) {
        Skin& sk = scene.skins[si];
        rfw_mat4 acc = mat4_identity();
        for (size_t k = 0; k < sk.joint_matrices.size(); k++) {
            const float ang = (si == 0 ? 0.35f : -0.25f) * std::sin(time * (1.0f + 0.3f * (float)si) + 0.4f * (float)k);
            const float piv[3] = {0.0f, 0.5f * (float)k, 0.0f};
            // local = T(piv) * Rz(ang) * T(-piv); accumulate down the chain
            const float c = std::cos(ang), sn = std::sin(ang);
            rfw_mat4 l = mat4_identity();
            l.m[0] = c; l.m[1] = sn; l.m[4] = -sn; l.m[5] = c;
            l.m[12] = piv[0] - (c * piv[0] - sn * piv[1]);
            l.m[13] = piv[1] - (sn * piv[0] + c * piv[1]);
            rfw_mat4 r;
            for (int col = 0; col < 4; col++)
                for (int row = 0; row < 4; row++) {
                    float v = 0.0f;
                    for (int i = 0; i < 4; i++) v += acc.m[i * 4 + row] * l.m[col * 4 + i];
                    r.m[col * 4 + row] = v;
                }
            acc = r;
            sk.joint_matrices[k] = acc;
        }
    }
    scene.skins_changed = true;
}

void build_skinned(Scene& scene, Camera3D& cam, uint32_t seed)
{
    (void)seed;
    Material wall; wall.color[0] = 0.7f; wall.color[1] = 0.7f; wall.color[2] = 0.65f; wall.roughness = 0.9f;
    Material skin_m; skin_m.color[0] = 0.85f; skin_m.color[1] = 0.45f; skin_m.color[2] = 0.3f; skin_m.roughness = 0.35f; skin_m.clearcoat = 0.4f;
    Material light; light.color[0] = 15.0f; light.color[1] = 14.0f; light.color[2] = 12.0f;
    const int m_wall = (int)scene.add_material(wall), m_skin = (int)scene.add_material(skin_m), m_light = (int)scene.add_material(light);
    Builder room;
    room.d.name = "room";
    room.quad(v3(-4, 0, -4), v3(-4, 0, 4), v3(4, 0, 4), v3(4, 0, -4), m_wall);
    room.quad(v3(-4, 0, 4), v3(-4, 4, 4), v3(4, 4, 4), v3(4, 0, 4), m_wall);
    room.quad(v3(-1, 3.9f, -1), v3(1, 3.9f, -1), v3(1, 3.9f, 1), v3(-1, 3.9f, 1), m_light);
    scene.add_instance(scene.add_mesh(Mesh3D::from(room.d)), mat4_identity());
    // the tube: radius 0.18, height 2, 4 joints at y = 0, 0.5, 1.0, 1.5; each vertex blends the two nearest joints
    Builder tube;
    tube.d.name = "tube";
    const float PI = 3.14159265358979323846f;
    tube.surface([&](float u, float v) { const float a = 2 * PI * u; return v3(0.18f * std::cos(a), 2.0f * v, 0.18f * std::sin(a)); }, 14, 16, m_skin, true);
    Mesh3D tm = Mesh3D::from(tube.d);
    tm.skin_data.resize(tm.vertices.size());
    for (size_t i = 0; i < tm.vertices.size(); i++) {
        const float y = tm.vertices[i].vertex.y, f = std::min(std::max(y / 0.5f, 0.0f), 2.9999f);
        const uint32_t j0 = (uint32_t)f;
        const float w1 = f - (float)j0;
        rfw_joint_data jd;
        std::memset(&jd, 0, sizeof(jd));
        jd.joint[0] = j0; jd.joint[1] = j0 + 1; jd.joint[2] = 0; jd.joint[3] = 0;
        jd.weight = rfw_vec4{1.0f - w1, w1, 0.0f, 0.0f};
        tm.skin_data[i] = jd;
    }
    const uint32_t mesh = scene.add_mesh(tm);
    for (int k = 0; k < 2; k++) {
        Skin sk;
        sk.inverse_bind_matrices.assign(4, mat4_identity());
        sk.joint_matrices.assign(4, mat4_identity());
        scene.skins.push_back(sk);
    }
    const size_t i0 = scene.add_instance(mesh, mat4_from_translation(-1.2f, 0.0f, 0.5f));
    const size_t i1 = scene.add_instance(mesh, mat4_from_translation(0.9f, 0.0f, 1.0f));
    scene.add_instance(mesh, mat4_from_translation(2.4f, 0.0f, -0.5f)); // bind pose, unskinned
    scene.instances_3d[mesh].skin_ids[i0] = 0;
    scene.instances_3d[mesh].skin_ids[i1] = 1;
    pose_skins(scene, 0.7f);
    scene.update_lights();
    cam = Camera3D();
    cam.pos[0] = 0.3f; cam.pos[1] = 1.6f; cam.pos[2] = -5.0f;
    const V3 d = normalize(v3(0.0f, -0.1f, 1.0f));
    cam.direction[0] = d.x; cam.direction[1] = d.y; cam.direction[2] = d.z;
    cam.fov = 55.0f;
    cam.aperture = 0.0f;
}

} // namespace rfw

// ==================================================================== C exports (for the Python tests / bench)
#define HOST_API extern "C" __attribute__((visibility("default")))
namespace {
struct HostScene {
    rfw::Scene scene;
    rfw::Camera3D cam;
    std::vector<rfw_device_material> dev_mats;
    uint32_t grid_mesh = 0, grid_nx = 0, grid_nz = 0;
    float grid_spacing = 1.0f;
    std::string error;
};
// Backend over a table of C function pointers with the rfw_hip_* signatures: lets the tests drive the product
// library and the oracle through the SAME synchronize_system.
struct rfwhost_backend_table {
    void* instance;
    int (*set_3d_mesh)(void*, uint32_t, const rfw_mesh_data_3d*);
    int (*unload_3d_meshes)(void*, const uint32_t*, uint32_t);
    int (*set_3d_instances)(void*, uint32_t, const rfw_instances_data_3d*);
    int (*set_materials)(void*, const rfw_device_material*, uint32_t, const uint32_t*);
    int (*synchronize)(void*);
    int (*set_point_lights)(void*, const rfw_point_light*, uint32_t, const uint32_t*);
    int (*set_spot_lights)(void*, const rfw_spot_light*, uint32_t, const uint32_t*);
    int (*set_area_lights)(void*, const rfw_area_light*, uint32_t, const uint32_t*);
    int (*set_directional_lights)(void*, const rfw_directional_light*, uint32_t, const uint32_t*);
    int (*set_textures)(void*, const rfw_texture_data*, uint32_t, const uint32_t*);
    int (*set_skybox)(void*, const rfw_texture_data*);
    int (*set_skins)(void*, const rfw_skin_data*, uint32_t, const uint32_t*);
};
struct TableBackend : rfw::Backend {
    rfwhost_backend_table t;
    int rc = 0;
    void acc(int r) { if (r != 0 && rc == 0) rc = r; }
    void set_2d_mesh(size_t, const void*, uint32_t, int32_t) override {}
    void set_2d_instances(size_t, const rfw_mat4*, uint32_t) override {}
    void set_3d_mesh(size_t id, const rfw_mesh_data_3d& d) override { acc(t.set_3d_mesh(t.instance, (uint32_t)id, &d)); }
    void unload_3d_meshes(const std::vector<size_t>& ids) override
    {
        std::vector<uint32_t> v(ids.begin(), ids.end());
        acc(t.unload_3d_meshes(t.instance, v.data(), (uint32_t)v.size()));
    }
    void set_3d_instances(size_t mesh, const rfw_instances_data_3d& d) override { acc(t.set_3d_instances(t.instance, (uint32_t)mesh, &d)); }
    void set_materials(const std::vector<rfw_device_material>& m, const std::vector<uint32_t>* ch) override
    {
        acc(t.set_materials(t.instance, m.data(), (uint32_t)m.size(), ch ? ch->data() : nullptr));
    }
    void set_textures(const std::vector<rfw_texture_data>& t, const std::vector<uint32_t>*) override
    {
        if (this->t.set_textures) acc(this->t.set_textures(this->t.instance, t.data(), (uint32_t)t.size(), nullptr));
    }
    void synchronize() override { acc(t.synchronize(t.instance)); }
    void render(const rfw_mat4&, const rfw_camera_view_3d&, uint32_t) override {}
    void resize(uint32_t, uint32_t, double) override {}
    void set_point_lights(const std::vector<rfw_point_light>& l, const std::vector<uint32_t>*) override
    {
        acc(t.set_point_lights(t.instance, l.data(), (uint32_t)l.size(), nullptr));
    }
    void set_spot_lights(const std::vector<rfw_spot_light>& l, const std::vector<uint32_t>*) override
    {
        acc(t.set_spot_lights(t.instance, l.data(), (uint32_t)l.size(), nullptr));
    }
    void set_area_lights(const std::vector<rfw_area_light>& l, const std::vector<uint32_t>*) override
    {
        acc(t.set_area_lights(t.instance, l.data(), (uint32_t)l.size(), nullptr));
    }
    void set_directional_lights(const std::vector<rfw_directional_light>& l, const std::vector<uint32_t>*) override
    {
        acc(t.set_directional_lights(t.instance, l.data(), (uint32_t)l.size(), nullptr));
    }
    void set_skybox(const rfw_texture_data& s) override
    {
        if (t.set_skybox) acc(t.set_skybox(t.instance, &s));
    }
    void set_skins(const std::vector<rfw_skin_data>& s, const std::vector<uint32_t>*) override
    {
        if (t.set_skins) acc(t.set_skins(t.instance, s.data(), (uint32_t)s.size(), nullptr));
    }
};
} // namespace

HOST_API void* rfwhost_scene_create() { return new HostScene(); }
HOST_API void rfwhost_scene_destroy(void* p) { delete (HostScene*)p; }
// kind: "cornell" | "atrium" (a = target triangles) | "soup" (a = triangles, b = instances) | "spheres" (adds a x b animated grid, spacing c)
HOST_API int rfwhost_build(void* p, const char* kind, uint32_t a, uint32_t b, float c, uint32_t seed)
{
    HostScene& h = *(HostScene*)p;
    const std::string k(kind);
    if (k == "cornell") rfw::build_cornell_box(h.scene, h.cam);
    else if (k == "atrium") rfw::build_atrium(h.scene, h.cam, a, seed, (int)b);
    else if (k == "soup") rfw::build_soup(h.scene, h.cam, a, b, seed);
    else if (k == "gallery") rfw::build_gallery(h.scene, h.cam, seed);
    else if (k == "skinned") rfw::build_skinned(h.scene, h.cam, seed);
    else if (k == "spheres") {
        rfw::add_sphere_grid(h.scene, a, b, c);
        h.grid_mesh = h.scene.meshes_3d.rbegin()->first;
        h.grid_nx = a; h.grid_nz = b; h.grid_spacing = c;
    } else return -1;
    return 0;
}
// glTF 2.0 file (.gltf or .glb) added to the scene; 0 on success, -1 with the message in rfwhost_last_error
HOST_API int rfwhost_load_gltf(void* p, const char* path, int use_camera)
{
    HostScene& h = *(HostScene*)p;
    if (!path) { h.error = "load_gltf: null path"; return -1; }
    try {
        std::string err;
        if (!rfw::load_gltf(path, h.scene, use_camera ? &h.cam : nullptr, err)) { h.error = err; return -1; }
    } catch (const std::exception& e) {
        h.error = std::string("load_gltf: ") + e.what();
        return -1;
    }
    return 0;
}
// Quad3D::new(normal, position, width, height, material) added as a mesh with one instance (examples/nphysics/src/main.rs:92 builds its
// ground this way); returns the mesh id
HOST_API int rfwhost_add_quad(void* p, const float* normal, const float* position, float width, float height, uint32_t material)
{
    HostScene& h = *(HostScene*)p;
    if (!normal || !position || material >= h.scene.materials.size()) return -1;
    const uint32_t mesh = h.scene.add_mesh(rfw::Mesh3D::from(rfw::make_quad(normal, position, width, height, material)));
    h.scene.add_instance(mesh, rfw::mat4_identity());
    h.scene.update_lights();
    return (int)mesh;
}
// Wavefront OBJ (+ its material libraries and textures) added to the scene as one mesh with one instance; returns the mesh id, -1 with the
// message in rfwhost_last_error
HOST_API int rfwhost_load_obj(void* p, const char* path)
{
    HostScene& h = *(HostScene*)p;
    if (!path) { h.error = "load_obj: null path"; return -1; }
    try {
        std::string err;
        uint32_t mesh = 0;
        if (!rfw::load_obj(path, h.scene, err, true, &mesh)) { h.error = err; return -1; }
        return (int)mesh;
    } catch (const std::exception& e) {
        h.error = std::string("load_obj: ") + e.what();
        return -1;
    }
}
// material `index` as the scene holds it: colour(4), specular(4), then metallic, subsurface, specular_f, roughness, specular_tint, anisotropic,
// sheen, sheen_tint, clearcoat, clearcoat_gloss, transmission, eta (20 floats) and the five texture ids (diffuse, normal,
// metallic-roughness, emissive, sheen)
HOST_API int rfwhost_material(void* p, uint32_t index, float* out20, int32_t* tex5)
{
    HostScene& h = *(HostScene*)p;
    if (index >= h.scene.materials.size()) return -1;
    const rfw::Material& m = h.scene.materials[index];
    if (out20) {
        std::memcpy(out20, m.color, 16); std::memcpy(out20 + 4, m.specular, 16);
        const float v[12] = {m.metallic, m.subsurface, m.specular_f, m.roughness, m.specular_tint, m.anisotropic, m.sheen, m.sheen_tint, m.clearcoat, m.clearcoat_gloss, m.transmission, m.eta};
        std::memcpy(out20 + 8, v, sizeof(v));
    }
    if (tex5) { tex5[0] = m.diffuse_tex; tex5[1] = m.normal_tex; tex5[2] = m.metallic_roughness_tex; tex5[3] = m.emissive_tex; tex5[4] = m.sheen_tex; }
    return 0;
}
// texture `index`: size of level 0; with `bgra_out` (w * h * 4 bytes) also its texels
HOST_API int rfwhost_texture(void* p, uint32_t index, uint32_t* w, uint32_t* hgt, uint8_t* bgra_out, uint64_t cap)
{
    HostScene& h = *(HostScene*)p;
    if (index >= h.scene.textures.size()) return -1;
    const rfw::Texture& t = h.scene.textures[index];
    if (w) *w = t.width;
    if (hgt) *hgt = t.height;
    const uint64_t n = (uint64_t)t.width * t.height * 4;
    if (bgra_out && cap >= n && t.bytes.size() >= n) std::memcpy(bgra_out, t.bytes.data(), n);
    return (int)h.scene.textures.size();
}
HOST_API int rfwhost_save_glb(void* p, const char* path)
{
    HostScene& h = *(HostScene*)p;
    if (!path) { h.error = "save_glb: null path"; return -1; }
    std::string err;
    if (!rfw::save_glb(path, h.scene, &h.cam, err)) { h.error = err; return -1; }
    return 0;
}
// Scene edits for the incremental-synchronize tests.  op 0: replace mesh `id` by displaced sphere number a (surface seeded by `seed`),
// quality q (4 = 5120 triangles); 1: remove mesh `id` and its instances; 2: add displaced sphere number a as a NEW mesh with one instance
// (returns its id); 3: recolour material `id` (r, g, b from seed's bytes, roughness a / 255) and mark only that material changed
HOST_API int rfwhost_edit(void* p, uint32_t op, uint32_t id, uint32_t a, uint32_t q, uint32_t seed)
{
    HostScene& h = *(HostScene*)p;
    if (op == 0) {
        if (h.scene.meshes_3d.find(id) == h.scene.meshes_3d.end()) return -1;
        const rfw::Mesh3D& old = h.scene.meshes_3d[id];
        const uint32_t mat = old.triangles.empty() ? 0u : (uint32_t)old.triangles[0].mat_id;
        h.scene.replace_mesh(id, rfw::Mesh3D::from(rfw::make_displaced_sphere((int)a, seed, (int)q, mat)));
        return (int)id;
    }
    if (op == 1) { h.scene.remove_mesh(id); return 0; }
    if (op == 2) {
        const uint32_t m = h.scene.add_mesh(rfw::Mesh3D::from(rfw::make_displaced_sphere((int)a, seed, (int)q, 0u)));
        h.scene.add_instance(m, rfw::mat4_identity());
        return (int)m;
    }
    if (op == 3) {
        if (id >= h.scene.materials.size()) return -1;
        rfw::Material m = h.scene.materials[id];
        m.color[0] = (float)(seed & 255u) / 255.0f; m.color[1] = (float)((seed >> 8) & 255u) / 255.0f; m.color[2] = (float)((seed >> 16) & 255u) / 255.0f;
        m.roughness = (float)a / 255.0f;
        h.scene.set_material(id, m);
        return 0;
    }
    if (op == 4) { // repaint texture `id`: same size, texels from `seed` (level 0; the mip chain is rebuilt); marks only this texture changed
        if (id >= h.scene.textures.size()) return -1;
        rfw::Texture t = h.scene.textures[id];
        const uint32_t levels = t.mip_levels;
        t.mip_levels = 1;
        t.bytes.resize((size_t)t.width * t.height * 4);
        uint32_t x = seed * 2654435761u + 1u;
        for (size_t i = 0; i < t.bytes.size(); i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; t.bytes[i] = (i & 3) == 3 ? 255 : (uint8_t)(x >> 11); }
        if (levels > 1) t.generate_mipmaps(levels);
        h.scene.set_texture(id, t);
        return 0;
    }
    return -1;
}
HOST_API const char* rfwhost_last_error(void* p) { return ((HostScene*)p)->error.c_str(); }
HOST_API int rfwhost_animate(void* p, float time)
{
    HostScene& h = *(HostScene*)p;
    if (h.grid_nx == 0) return -1;
    rfw::animate_sphere_grid(h.scene, h.grid_mesh, h.grid_nx, h.grid_nz, h.grid_spacing, time);
    return 0;
}
HOST_API int rfwhost_pose(void* p, float time)
{
    HostScene& h = *(HostScene*)p;
    if (h.scene.skins.empty()) return -1;
    rfw::pose_skins(h.scene, time);
    return 0;
}
// crates/rfw-scene/src/lib.rs:685-687 set_animations_time: the active animation of every loaded glTF graph at `time` seconds (looping);
// returns the number of animations the scene holds (0: nothing moved)
HOST_API int rfwhost_set_animation_time(void* p, double time)
{
    HostScene& h = *(HostScene*)p;
    int n = 0;
    for (const rfw::NodeGraph& g : h.scene.graphs) n += (int)g.animations.size();
    h.scene.set_animations_time(time);
    return n;
}
// GraphHandle::get_transform() of loaded graph `graph` (in load order): translation, rotation quaternion (x, y, z, w), scale
HOST_API int rfwhost_set_graph_transform(void* p, uint32_t graph, const double* t, const double* q, const double* s)
{
    HostScene& h = *(HostScene*)p;
    if (graph >= h.scene.graphs.size() || !t || !q || !s) return -1;
    h.scene.graphs[graph].set_root_transform(h.scene, t, q, s);
    return 0;
}
// Scene::instantiate_graph: a second instance of loaded graph `graph` over the same meshes; returns the new graph's index, -1: no such graph
HOST_API int rfwhost_instantiate_graph(void* p, uint32_t graph)
{
    HostScene& h = *(HostScene*)p;
    if (graph >= h.scene.graphs.size()) return -1;
    return (int)h.scene.instantiate_graph(graph);
}
HOST_API int rfwhost_graph_count(void* p) { return (int)((HostScene*)p)->scene.graphs.size(); }
// duration (seconds) and channel count of animation `index` counted over the loaded graphs; -1: no such animation
HOST_API int rfwhost_animation_info(void* p, uint32_t index, double* duration, uint32_t* channels)
{
    HostScene& h = *(HostScene*)p;
    for (const rfw::NodeGraph& g : h.scene.graphs) {
        if (index < g.animations.size()) {
            if (duration) *duration = g.animations[index].duration;
            if (channels) *channels = (uint32_t)g.animations[index].channels.size();
            return 0;
        }
        index -= (uint32_t)g.animations.size();
    }
    return -1;
}
// joint matrices of skin `skin` (16 floats each, column-major) -> out; returns the number of joints (copies at most `cap` of them), -1: no such skin
HOST_API int rfwhost_skin_matrices(void* p, uint32_t skin, float* out, uint32_t cap)
{
    HostScene& h = *(HostScene*)p;
    if (skin >= h.scene.skins.size()) return -1;
    const std::vector<rfw_mat4>& jm = h.scene.skins[skin].joint_matrices;
    for (size_t j = 0; j < jm.size() && j < cap && out; j++) std::memcpy(out + 16 * j, &jm[j], 16 * sizeof(float));
    return (int)jm.size();
}
HOST_API int rfwhost_instance_matrix(void* p, uint32_t mesh, uint32_t slot, float* out16)
{
    HostScene& h = *(HostScene*)p;
    auto it = h.scene.instances_3d.find(mesh);
    if (it == h.scene.instances_3d.end() || slot >= it->second.matrices.size() || !out16) return -1;
    std::memcpy(out16, &it->second.matrices[slot], 16 * sizeof(float));
    return it->second.skin_ids.size() > slot ? it->second.skin_ids[slot] + 1 : 0; // 0: unskinned, else skin id + 1
}
// PNG or JPEG bytes -> RGBA8; returns 0 and the size, -1 with the reason in *err_out (static storage, valid until the next call on this thread)
HOST_API int rfwhost_decode_image(const uint8_t* data, uint64_t n, uint32_t* w, uint32_t* hgt, uint8_t* rgba_out, uint64_t cap, const char** err_out)
{
    static thread_local std::string err;
    std::vector<uint8_t> rgba;
    uint32_t ww = 0, hh = 0;
    err.clear();
    bool ok = false;
    try {
        ok = data && rfw::decode_image(data, (size_t)n, ww, hh, rgba, err);
    } catch (const std::exception& e) { // an allocation the file's header asked for
        err = std::string("image: ") + e.what();
    }
    if (!ok) {
        if (err_out) *err_out = err.c_str();
        return -1;
    }
    if (w) *w = ww;
    if (hgt) *hgt = hh;
    if (rgba_out && cap >= rgba.size()) std::memcpy(rgba_out, rgba.data(), rgba.size());
    return 0;
}
HOST_API int rfwhost_set_camera(void* p, const float* pos, const float* dir, float fov, float aperture, float aspect)
{
    HostScene& h = *(HostScene*)p;
    for (int i = 0; i < 3; i++) { h.cam.pos[i] = pos[i]; h.cam.direction[i] = dir[i]; }
    h.cam.fov = fov; h.cam.aperture = aperture; h.cam.aspect_ratio = aspect;
    return 0;
}
// op 0: translate_relative(a), 1: translate_target(a), 2: look_at(a, b); pos_dir_out (6 floats, optional): the camera afterwards
HOST_API int rfwhost_camera_move(void* p, int op, const float* a, const float* b, float* pos_dir_out)
{
    HostScene& h = *(HostScene*)p;
    if (!a || (op == 2 && !b)) return -1;
    if (op == 0) h.cam.translate_relative(a);
    else if (op == 1) h.cam.translate_target(a);
    else if (op == 2) h.cam.look_at(a, b);
    else return -1;
    if (pos_dir_out) { std::memcpy(pos_dir_out, h.cam.pos, 12); std::memcpy(pos_dir_out + 3, h.cam.direction, 12); }
    return 0;
}
HOST_API int rfwhost_set_aspect(void* p, float aspect) { ((HostScene*)p)->cam.aspect_ratio = aspect; return 0; }
HOST_API int rfwhost_camera_view(void* p, uint32_t w, uint32_t h, rfw_camera_view_3d* out) { *out = ((HostScene*)p)->cam.get_view(w, h); return 0; }
HOST_API int rfwhost_mark_all_changed(void* p)
{
    HostScene& h = *(HostScene*)p;
    for (auto& kv : h.scene.meshes_3d) h.scene.mesh_changed[kv.first] = true;
    for (auto& kv : h.scene.instances_3d) h.scene.instances_changed[kv.first] = true;
    h.scene.materials_changed = true;
    h.scene.material_changed_bits.clear();
    h.scene.lights_changed = true;
    h.scene.textures_changed = !h.scene.textures.empty();
    h.scene.texture_changed_bits.clear();
    h.scene.skybox_changed = h.scene.skybox.width != 0;
    h.scene.skins_changed = !h.scene.skins.empty();
    return 0;
}
// runs rfw::synchronize_system against a table of C entry points (rfw_hip_* or orc_*)
HOST_API int rfwhost_synchronize(void* p, const rfwhost_backend_table* table)
{
    HostScene& h = *(HostScene*)p;
    TableBackend b;
    b.t = *table;
    rfw::synchronize_system(h.scene, b);
    return b.rc;
}
HOST_API uint64_t rfwhost_triangle_count(void* p) { return ((HostScene*)p)->scene.triangle_count(); }
HOST_API uint32_t rfwhost_counts(void* p, uint32_t what)
{
    HostScene& h = *(HostScene*)p;
    switch (what) {
    case 0: return (uint32_t)h.scene.meshes_3d.size();
    case 1: { uint32_t n = 0; for (auto& kv : h.scene.instances_3d) n += (uint32_t)kv.second.matrices.size(); return n; }
    case 2: return (uint32_t)h.scene.materials.size();
    case 3: return (uint32_t)h.scene.area_lights.size();
    case 4: return (uint32_t)h.scene.point_lights.size();
    case 5: return (uint32_t)h.scene.spot_lights.size();
    case 6: return (uint32_t)h.scene.directional_lights.size();
    default: return 0;
    }
}
HOST_API int rfwhost_mesh_data(void* p, uint32_t id, rfw_mesh_data_3d* out)
{
    HostScene& h = *(HostScene*)p;
    auto it = h.scene.meshes_3d.find(id);
    if (it == h.scene.meshes_3d.end()) return -1;
    *out = it->second.as_data();
    return 0;
}
HOST_API int rfwhost_into_device_material(const float* color, const float* params16, rfw_device_material* out)
{
    rfw::Material m;
    for (int i = 0; i < 4; i++) m.color[i] = color[i];
    m.metallic = params16[0]; m.subsurface = params16[1]; m.specular_f = params16[2]; m.roughness = params16[3];
    m.specular_tint = params16[4]; m.anisotropic = params16[5]; m.sheen = params16[6]; m.sheen_tint = params16[7];
    m.clearcoat = params16[8]; m.clearcoat_gloss = params16[9]; m.transmission = params16[10]; m.eta = params16[11];
    m.custom0 = params16[12]; m.custom1 = params16[13]; m.custom2 = params16[14]; m.custom3 = params16[15];
    *out = rfw::into_device_material(m);
    return 0;
}
