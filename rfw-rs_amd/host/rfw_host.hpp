/*
 * rfw_host.hpp — C++ host side above the C ABI.
 *
 * The reference's host code is Rust; this image has no Rust toolchain, so the host
 * side is written in C++ and mirrors the reference's operator/plugin interface for
 * the path, name for name:
 *   rfw::Backend            <- trait rfw_backend::Backend   (crates/rfw-backend/src/lib.rs:35-82)
 *   rfw::HipBackend::init   <- FromWindowHandle::init        (crates/rfw-backend/src/lib.rs:26-33)
 *   rfw::synchronize_system <- rfw/src/system/mod.rs:19-206  (push changed meshes/instances/
 *                              materials/lights, then synchronize())
 *   rfw::render_system      <- rfw/src/lib.rs:411-430
 * plus the small part of rfw-scene that defines the INPUTS the backend sees
 * (SURVEY.md §8f rank 1): Mesh3D::from(MeshDescriptor), into_device_material,
 * Scene::update_lights, Camera3D::get_view.  The Rust shim a maintainer would add to
 * the reference instead of this file is in INTEGRATION.md.
 */
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rfw_hip.h"

namespace rfw {

// ---- the plugin interface (crates/rfw-backend/src/lib.rs:35-82), same names and argument meaning ----
struct Backend {
    virtual ~Backend() {}
    virtual void set_2d_mesh(size_t id, const void* vertices, uint32_t num_vertices, int32_t tex_id) = 0;
    virtual void set_2d_instances(size_t mesh, const rfw_mat4* matrices, uint32_t n) = 0;
    virtual void set_3d_mesh(size_t id, const rfw_mesh_data_3d& data) = 0;
    virtual void unload_3d_meshes(const std::vector<size_t>& ids) = 0;
    virtual void set_3d_instances(size_t mesh, const rfw_instances_data_3d& instances) = 0;
    virtual void set_materials(const std::vector<rfw_device_material>& materials, const std::vector<uint32_t>* changed) = 0;
    virtual void set_textures(const std::vector<rfw_texture_data>& textures, const std::vector<uint32_t>* changed) = 0;
    virtual void synchronize() = 0;
    virtual void render(const rfw_mat4& view_2d, const rfw_camera_view_3d& view_3d, uint32_t mode) = 0;
    virtual void resize(uint32_t width, uint32_t height, double scale_factor) = 0;
    virtual void set_point_lights(const std::vector<rfw_point_light>& lights, const std::vector<uint32_t>* changed) = 0;
    virtual void set_spot_lights(const std::vector<rfw_spot_light>& lights, const std::vector<uint32_t>* changed) = 0;
    virtual void set_area_lights(const std::vector<rfw_area_light>& lights, const std::vector<uint32_t>* changed) = 0;
    virtual void set_directional_lights(const std::vector<rfw_directional_light>& lights, const std::vector<uint32_t>* changed) = 0;
    virtual void set_skybox(const rfw_texture_data& skybox) = 0;
    virtual void set_skins(const std::vector<rfw_skin_data>& skins, const std::vector<uint32_t>* changed) = 0;
};

// The MI355X backend: every trait method forwards to the C entry point of the same name.
// Trait methods return (); failures surface the way the reference's backends do (panic) — here a C++ exception.
class HipBackend : public Backend {
public:
    // FromWindowHandle::init(window, width, height, scale) — headless: the window handle is ignored.
    static HipBackend* init(uint32_t width, uint32_t height, double scale, const rfw_hip_options* options = nullptr)
    {
        if (rfw_hip_abi_version() != RFW_HIP_ABI_VERSION) throw std::runtime_error("librfw_hip.so was built from another version of include/rfw_hip.h");
        void* inst = rfw_hip_create(width, height, scale, options);
        if (!inst) throw std::runtime_error(std::string("rfw_hip_create: ") + rfw_hip_last_error(nullptr));
        return new HipBackend(inst);
    }
    ~HipBackend() override { rfw_hip_destroy(inst_); }
    void* raw() const { return inst_; }

    void set_2d_mesh(size_t id, const void* v, uint32_t n, int32_t tex) override { check(rfw_hip_set_2d_mesh(inst_, (uint32_t)id, v, n, tex)); }
    void set_2d_instances(size_t mesh, const rfw_mat4* m, uint32_t n) override { check(rfw_hip_set_2d_instances(inst_, (uint32_t)mesh, m, n)); }
    void set_3d_mesh(size_t id, const rfw_mesh_data_3d& d) override { check(rfw_hip_set_3d_mesh(inst_, (uint32_t)id, &d)); }
    void unload_3d_meshes(const std::vector<size_t>& ids) override
    {
        std::vector<uint32_t> v(ids.begin(), ids.end());
        check(rfw_hip_unload_3d_meshes(inst_, v.data(), (uint32_t)v.size()));
    }
    void set_3d_instances(size_t mesh, const rfw_instances_data_3d& d) override { check(rfw_hip_set_3d_instances(inst_, (uint32_t)mesh, &d)); }
    void set_materials(const std::vector<rfw_device_material>& m, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_materials(inst_, m.data(), (uint32_t)m.size(), ch ? ch->data() : nullptr));
    }
    void set_textures(const std::vector<rfw_texture_data>& t, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_textures(inst_, t.data(), (uint32_t)t.size(), ch ? ch->data() : nullptr));
    }
    void synchronize() override { check(rfw_hip_synchronize(inst_)); }
    void render(const rfw_mat4& v2, const rfw_camera_view_3d& v3, uint32_t mode) override { check(rfw_hip_render(inst_, &v2, &v3, mode)); }
    void resize(uint32_t w, uint32_t h, double s) override { check(rfw_hip_resize(inst_, w, h, s)); }
    void set_point_lights(const std::vector<rfw_point_light>& l, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_point_lights(inst_, l.data(), (uint32_t)l.size(), ch ? ch->data() : nullptr));
    }
    void set_spot_lights(const std::vector<rfw_spot_light>& l, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_spot_lights(inst_, l.data(), (uint32_t)l.size(), ch ? ch->data() : nullptr));
    }
    void set_area_lights(const std::vector<rfw_area_light>& l, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_area_lights(inst_, l.data(), (uint32_t)l.size(), ch ? ch->data() : nullptr));
    }
    void set_directional_lights(const std::vector<rfw_directional_light>& l, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_directional_lights(inst_, l.data(), (uint32_t)l.size(), ch ? ch->data() : nullptr));
    }
    void set_skybox(const rfw_texture_data& s) override { check(rfw_hip_set_skybox(inst_, &s)); }
    void set_skins(const std::vector<rfw_skin_data>& s, const std::vector<uint32_t>* ch) override
    {
        check(rfw_hip_set_skins(inst_, s.data(), (uint32_t)s.size(), ch ? ch->data() : nullptr));
    }

private:
    explicit HipBackend(void* inst) : inst_(inst) {}
    void check(int rc) const
    {
        if (rc != RFW_HIP_OK) throw std::runtime_error(std::string("rfw_hip: ") + rfw_hip_last_error(inst_));
    }
    void* inst_;
};

// ---- scene-side inputs (the part of rfw-scene that feeds the boundary) ----

// l3d::mat::Material as consumed by into_device_material (crates/rfw-scene/src/material/list.rs:755-814)
struct Material {
    float color[4] = {1, 1, 1, 1};
    float absorption[4] = {0, 0, 0, 0};
    float specular[4] = {1, 1, 1, 1};
    float metallic = 0, subsurface = 0, specular_f = 0.5f, roughness = 0.5f;
    float specular_tint = 0, anisotropic = 0, sheen = 0, sheen_tint = 0;
    float clearcoat = 0, clearcoat_gloss = 1, transmission = 0, eta = 1;
    float custom0 = 0, custom1 = 0, custom2 = 0, custom3 = 0;
    int diffuse_tex = -1, normal_tex = -1, metallic_roughness_tex = -1, emissive_tex = -1, sheen_tex = -1;
};
rfw_device_material into_device_material(const Material& m);
// crates/rfw-scene/src/material/list.rs:492-515 light_flags(): any(color.rgb > 1)
bool is_emissive(const Material& m);

// l3d MeshDescriptor fields used by Mesh3D::from (crates/rfw-scene/src/objects_3d/mod.rs:673-895): non-indexed, 3 vertices per triangle
struct MeshDescriptor {
    std::vector<rfw_vec4> vertices;
    std::vector<rfw_vec3> normals;   // all-zero first normal => generated
    std::vector<rfw_vec2> uvs;
    std::vector<rfw_vec4> tangents;
    std::vector<int32_t> material_ids; // per vertex
    std::string name;
};

struct Mesh3D {
    std::string name;
    std::vector<rfw_vertex_3d> vertices;
    std::vector<rfw_rt_triangle> triangles;
    std::vector<rfw_vertex_mesh> ranges;
    std::vector<rfw_joint_data> skin_data; // per vertex (3 per triangle), empty = not skinnable
    std::vector<uint32_t> materials;
    rfw_aabb bounds;
    uint32_t flags = RFW_MESH_SHADOW_CASTER | RFW_MESH_ALLOW_SKINNING;
    static Mesh3D from(const MeshDescriptor& desc);
    rfw_mesh_data_3d as_data() const;
};

// crates/rfw-scene/src/instances_3d.rs:11-165
struct InstanceList3D {
    std::vector<rfw_mat4> matrices;
    std::vector<int32_t> skin_ids;
    std::vector<uint32_t> flags;
    size_t allocate(const rfw_mat4& m);
    void make_invalid(size_t slot); // zero matrix (instances_3d.rs:79-86)
};

// crates/rfw-scene/src/camera/mod.rs:29-115
struct Camera3D {
    float pos[3] = {0, 0, 0};
    float direction[3] = {0, 0, 1};
    float fov = 40.0f, aspect_ratio = 1.0f, aperture = 0.0001f, focal_distance = 1.0f, near_plane = 1e-2f, far_plane = 1e5f;
    float speed = 1.0f;
    rfw_camera_view_3d get_view(uint32_t width, uint32_t height) const;
    // camera/mod.rs:164-186: the moves the reference's examples make from their key handlers (examples/animated/src/main.rs:146-195)
    void translate_relative(const float delta[3]); // along the camera's right / up / forward, scaled by `speed`
    void translate_target(const float delta[3]);   // turns: direction += delta in the camera's frame, normalised
    void look_at(const float origin[3], const float target[3]);
};

// crates/rfw-backend/src/structs.rs:69-121 TextureData: 4 bytes per texel, mip levels concatenated (level i is (w >> i) x (h >> i))
struct Texture {
    uint32_t width = 0, height = 0, mip_levels = 1;
    uint32_t format = RFW_FORMAT_BGRA8;
    std::vector<uint8_t> bytes;
    rfw_texture_data as_data() const;
    // box-filtered mip chain appended to level 0 (what l3d's Texture::generate_mipmaps hands the trait)
    void generate_mipmaps(uint32_t levels);
};

// crates/rfw-backend/src/structs.rs:6-11 SkinData (the joint matrices are what SkinnedTriangles3D::apply consumes)
struct Skin {
    std::vector<rfw_mat4> inverse_bind_matrices, joint_matrices;
};

// The node graph of a loaded glTF document, kept so that its animations can run (crates/rfw-scene/src/graph/mod.rs:100-115 Node,
// :338-344 NodeGraph, :835-840 Skin::joint_nodes; channels and samplers are l3d's Animation, sampled as glTF 2.0 section 3.11 says).
struct GraphNode {
    double translation[3] = {0, 0, 0}, rotation[4] = {0, 0, 0, 1}, scale[3] = {1, 1, 1};
    bool has_matrix = false;       // a node given as a matrix is not animated (glTF: animated nodes use TRS)
    double matrix[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // combined_matrix
    std::vector<uint32_t> children;
    int64_t mesh = -1, instance = -1; // scene mesh id and the instance slot this node owns (-1: none)
    int32_t skin = -1;                // scene skin id worn by that instance
};
struct AnimationSampler {
    std::vector<double> times, values; // values: components per key (x 3 for cubic splines: in-tangent, value, out-tangent)
    int interpolation = 0;             // 0 LINEAR, 1 STEP, 2 CUBICSPLINE
};
struct AnimationChannel {
    uint32_t node = 0, sampler = 0;
    int path = 0; // 0 translation, 1 rotation, 2 scale (morph-target weights are not animated: graph/mod.rs:577 "TODO: Morph animations")
};
struct Animation {
    std::string name;
    std::vector<AnimationSampler> samplers;
    std::vector<AnimationChannel> channels;
    double duration = 0.0; // the largest key time
};
struct Scene;
struct NodeGraph {
    std::vector<GraphNode> nodes;
    std::vector<uint32_t> order;                       // nodes reachable from the scene's roots, parents first
    std::vector<uint32_t> parent_of;                   // per node: parent index or UINT32_MAX
    std::vector<std::pair<int32_t, std::vector<int64_t>>> skins; // (scene skin id, joint node per joint; -1: none)
    std::vector<Animation> animations;
    int active_animation = 0;                          // graph/mod.rs:643-647 set_active_animation
    // the transform of the whole graph, what GraphHandle::get_transform() edits (graph/mod.rs:127-147; examples/animated/src/main.rs:84-103
    // places two CesiumMan graphs this way): parent of the document's root nodes.  Cameras and punctual lights of the file were placed at
    // load time and do not follow it.
    double root[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    void set_root_transform(Scene& scene, const double translation[3], const double rotation_xyzw[4], const double scale[3]);
    // graph/mod.rs:636-641 update_animation + :477-507 update: channel values at `time` (wrapped into the animation's duration) -> node
    // TRS -> combined matrices -> instance matrices and joint matrices of `scene`, which are marked changed
    void set_animation_time(Scene& scene, double time);
    void update(Scene& scene); // combined matrices from the nodes' current TRS -> instances and skins
};

struct Scene {
    std::vector<NodeGraph> graphs;
    // crates/rfw-scene/src/lib.rs:685-687 set_animations_time: every loaded graph's active animation
    void set_animations_time(double time);
    // Scene::add_3d(&SceneDescriptor) a second time (examples/animated/src/main.rs:84-103 adds the CesiumMan descriptor twice): a new graph
    // over the SAME meshes — new instances, new skins (two instances of one mesh may wear different skins), its own transform and clock.
    // Returns the new graph's index.
    size_t instantiate_graph(size_t graph);
    std::vector<Skin> skins;
    bool skins_changed = false;
    std::vector<Texture> textures;
    Texture skybox;
    bool textures_changed = false, skybox_changed = false;
    std::map<uint32_t, Mesh3D> meshes_3d;
    std::map<uint32_t, InstanceList3D> instances_3d;
    std::vector<Material> materials;
    std::vector<rfw_area_light> area_lights;
    std::vector<rfw_point_light> point_lights;
    std::vector<rfw_spot_light> spot_lights;
    std::vector<rfw_directional_light> directional_lights;
    // change tracking (rfw-utils TrackedStorage/FlaggedStorage bits)
    std::map<uint32_t, bool> mesh_changed, instances_changed;
    bool materials_changed = true, lights_changed = true;
    // which materials changed since the last synchronize_system (the reference's TrackedStorage bits, crates/rfw-scene/src/material/list.rs);
    // empty = all of them (a new scene, a resized list)
    std::vector<uint32_t> material_changed_bits;
    std::vector<uint32_t> texture_changed_bits; // like material_changed_bits: which textures changed since the last synchronize_system; empty = all
    std::vector<uint32_t> removed_meshes; // unloaded since the last synchronize_system (rfw/src/system/mod.rs: unload_3d_meshes)

    uint32_t add_material(const Material& m);
    uint32_t add_mesh(const Mesh3D& m);
    void replace_mesh(uint32_t id, const Mesh3D& m);   // same id, new geometry (set_3d_mesh again at the next synchronize_system)
    void remove_mesh(uint32_t id);                      // the mesh and its instances
    void set_material(uint32_t index, const Material& m); // marks only this material changed
    void set_texture(uint32_t index, const Texture& t);   // marks only this texture changed (set_textures then carries a `changed` bit slice)
    size_t add_instance(uint32_t mesh, const rfw_mat4& m);
    void set_matrix(uint32_t mesh, size_t slot, const rfw_mat4& m);
    // crates/rfw-scene/src/lib.rs:575-648
    void update_lights();
    std::vector<rfw_device_material> device_materials() const;
    uint64_t triangle_count() const;
};

rfw_mat4 mat4_identity();
rfw_mat4 mat4_from_translation(float x, float y, float z);
rfw_mat4 mat4_from_scale_translation(float s, float x, float y, float z);
rfw_mat4 mat4_from_trs(const float t[3], const float axis[3], float angle, float scale);

// rfw/src/system/mod.rs:19-206
void synchronize_system(Scene& scene, Backend& renderer);
// rfw/src/lib.rs:411-430
void render_system(const Camera3D& camera, uint32_t width, uint32_t height, Backend& renderer);

// ---- synthetic scenes standing in for the assets the reference does not ship ----
MeshDescriptor make_quad(const float normal[3], const float position[3], float width, float height, uint32_t mat_id); // objects_3d/quad.rs:19-75 Quad3D
MeshDescriptor make_icosphere(int quality, uint32_t mat_id);                       // objects_3d/sphere.rs:365-519
void build_cornell_box(Scene& scene, Camera3D& cam);                               // C1
// separate_spheres: C4's 64 displaced icospheres as 64 meshes with one instance each (65 meshes in all) instead of one baked mesh
void build_atrium(Scene& scene, Camera3D& cam, uint32_t target_triangles, uint32_t seed, int sphere_meshes = 0); // C2 ("Sponza-class") / C4; sphere_meshes: 0 one baked sphere mesh (2 meshes), 1 = 64 sphere meshes (65), 2 = everything in ONE mesh
// one of C4's displaced icospheres (k = 0..63): the 5120-triangle sphere at its place in the atrium, surface noise seeded by `seed`; quality 4 = 5120 triangles
MeshDescriptor make_displaced_sphere(int k, uint32_t seed, int quality, uint32_t mat_id);
void add_sphere_grid(Scene& scene, uint32_t nx, uint32_t nz, float spacing);       // C3: instances of a 320-tri icosphere
void animate_sphere_grid(Scene& scene, uint32_t mesh, uint32_t nx, uint32_t nz, float spacing, float time); // examples/animated/src/main.rs:197-219
void build_soup(Scene& scene, Camera3D& cam, uint32_t triangles, uint32_t instances, uint32_t seed); // random soups for BVH equivalence tests
void build_gallery(Scene& scene, Camera3D& cam, uint32_t seed);
// a skinned tube (4 joints) instanced twice with two different skins + one unskinned instance of the same mesh, in a lit room;
// pose_skins() bends the joints as a function of time (the graph/animation system of rfw-scene is what does this in the reference)
void build_skinned(Scene& scene, Camera3D& cam, uint32_t seed);
// glTF 2.0 (.gltf / .glb) -> meshes, materials, instances, skins (gltf.cpp; crates/rfw-scene/src/loaders/gltf.rs:26-90 via l3d)
bool load_gltf(const std::string& path, Scene& scene, Camera3D* cam, std::string& err);
// obj.cpp: Wavefront OBJ + MTL -> ONE mesh (all models of the file, de-indexed) and its materials, as the reference's ObjLoader does
// (crates/rfw-scene/src/loaders/obj.rs:26-252); add_instance: also place it once at the identity (scene.add_3d(&mesh))
bool load_obj(const std::string& path, Scene& scene, std::string& err, bool add_instance = true, uint32_t* mesh_out = nullptr);
bool decode_tga(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err);
// jpeg.cpp: JPEG (sequential or progressive) -> RGBA8 (the reference's CesiumMan sample carries a JPEG texture); gltf.cpp: PNG -> RGBA8
bool decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err);
bool decode_image(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err);
// gltf_export.cpp: the scene (static meshes, instances, materials, punctual lights, camera) as a binary glTF 2.0 file
bool save_glb(const std::string& path, const Scene& scene, const Camera3D* cam, std::string& err);
void pose_skins(Scene& scene, float time); // textured walls (diffuse + normal maps), an emissive-mapped panel, open sky with a lat-long skybox

} // namespace rfw
