/*
 * rfw_pod.h — plain-old-data wire structs of the rfw Backend boundary, as C.
 *
 * Every struct here is the byte-for-byte C mirror of a `#[repr(C)]` Rust type in
 * the reference crate `rfw-backend` (citations are relative to /root/reference).
 * The Rust side passes `&[T]` slices of these types; the C side receives
 * `const T*` + count.  Sizes/offsets are pinned by static asserts below — they
 * restate the only reference test that touches this boundary
 * (backends/metal/src/lib.rs:270-348 `test_layout`, size_of Rust == size_of C)
 * and the layout table of SURVEY.md Appendix A (glam x86-64/SSE: Vec3 12/4,
 * Vec4 16/16, Mat4 64/16).
 */
#ifndef RFW_POD_H
#define RFW_POD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RFW_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RFW_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
#define RFW_ALIGN16 __attribute__((aligned(16)))

/* glam::Vec2 / Vec3 / Vec4 / Mat4 (crates/rfw-math/src/lib.rs:1-33 re-exports glam). */
typedef struct { float x, y; } rfw_vec2;
typedef struct { float x, y, z; } rfw_vec3;
typedef struct RFW_ALIGN16 { float x, y, z, w; } rfw_vec4;
/* column-major: c[i] is column i (glam x_axis, y_axis, z_axis, w_axis). */
typedef struct RFW_ALIGN16 { float m[16]; } rfw_mat4;

/* rtbvh 0.6 `Aabb` — field order visible at crates/rfw-scene/src/camera/frustrum.rs:264-269. */
typedef struct {
    float min[3];
    int32_t extra1;
    float max[3];
    int32_t extra2;
} rfw_aabb;

/* crates/rfw-backend/src/structs.rs:879-918 (GLSL mirror: backends/gpu-rt/shaders/structs.glsl:67-108). */
typedef struct RFW_ALIGN16 {
    rfw_vec3 vertex0; float u0;
    rfw_vec3 vertex1; float u1;
    rfw_vec3 vertex2; float u2;
    rfw_vec3 normal;  float v0;
    rfw_vec3 n0;      float v1;
    rfw_vec3 n1;      float v2;
    rfw_vec3 n2;      int32_t id;
    rfw_vec4 tangent0;
    rfw_vec4 tangent1;
    rfw_vec4 tangent2;
    int32_t light_id;
    int32_t mat_id;
    float lod;
    float area;
} rfw_rt_triangle;

/* crates/rfw-backend/src/structs.rs:251-267 */
typedef struct RFW_ALIGN16 {
    rfw_vec4 vertex;
    rfw_vec3 normal;
    uint32_t mat_id;
    rfw_vec2 uv;
    float pad0, pad1;
    rfw_vec4 tangent;
} rfw_vertex_3d;

/* crates/rfw-backend/src/structs.rs:269-275 */
typedef struct RFW_ALIGN16 {
    uint32_t joint[4];
    rfw_vec4 weight;
} rfw_joint_data;

/* crates/rfw-backend/src/structs.rs:306-315 — `first`/`last` are VERTEX indices (3 x triangle index). */
typedef struct {
    rfw_aabb bounds;
    uint32_t first;
    uint32_t last;
    uint32_t mat_id;
    uint32_t padding;
} rfw_vertex_mesh;

/* crates/rfw-backend/src/structs.rs:369-394; parameter packing: crates/rfw-scene/src/material/list.rs:755-783 */
typedef struct {
    float color[4];
    float absorption[4];
    float specular[4];
    uint32_t parameters[4];
    uint32_t flags;
    int32_t diffuse_map;
    int32_t normal_map;
    int32_t metallic_roughness_map;
    int32_t emissive_map;
    int32_t sheen_map;
    int32_t _dummy[2];
} rfw_device_material;

/* crates/rfw-backend/src/structs.rs:484-515 */
typedef struct RFW_ALIGN16 {
    rfw_vec3 pos;
    rfw_vec3 right;
    rfw_vec3 up;
    rfw_vec3 p1;
    rfw_vec3 direction;
    float lens_size;
    float spread_angle;
    float epsilon;
    float inv_width;
    float inv_height;
    float near_plane;
    float far_plane;
    float aspect_ratio;
    float fov;
    rfw_vec4 custom0;
    rfw_vec4 custom1;
} rfw_camera_view_3d;

/* crates/rfw-backend/src/lights.rs:6-30 */
typedef struct {
    rfw_vec3 position; float energy;
    rfw_vec3 normal;   float area;
    rfw_vec3 vertex0;  int32_t inst_idx;
    rfw_vec3 vertex1;  int32_t mesh_id;
    rfw_vec3 radiance; int32_t _dummy1;
    rfw_vec3 vertex2;  int32_t _dummy2;
} rfw_area_light;

/* crates/rfw-backend/src/lights.rs:100-108 (Rust adds align(32); stride is 32 either way) */
typedef struct {
    rfw_vec3 position; float energy;
    rfw_vec3 radiance; float _dummy;
} rfw_point_light;

/* crates/rfw-backend/src/lights.rs:199-209 */
typedef struct {
    rfw_vec3 position;  float cos_inner;
    rfw_vec3 radiance;  float cos_outer;
    rfw_vec3 direction; float energy;
} rfw_spot_light;

/* crates/rfw-backend/src/lights.rs:293-301 */
typedef struct {
    rfw_vec3 direction; float energy;
    rfw_vec3 radiance;  float _dummy;
} rfw_directional_light;

/* crates/rfw-backend/src/structs.rs:28-34, 317-324 (repr(transparent) u32 bitflags) */
enum { RFW_INSTANCE_TRANSFORMED = 1u };
enum { RFW_MESH_SHADOW_CASTER = 1u, RFW_MESH_ALLOW_SKINNING = 2u };
/* crates/rfw-backend/src/structs.rs:62-67 */
enum { RFW_FORMAT_BGRA8 = 0u, RFW_FORMAT_RGBA8 = 1u };
/* crates/rfw-backend/src/lib.rs:9-18 */
enum {
    RFW_RENDER_DEFAULT = 0, RFW_RENDER_NORMAL = 1, RFW_RENDER_ALBEDO = 2, RFW_RENDER_GBUFFER = 3,
    RFW_RENDER_SCREEN_SPACE = 4, RFW_RENDER_SSAO = 5, RFW_RENDER_FILTERED_SSAO = 6
};
/* crates/rfw-scene/src/material/mod.rs:27-34 */
enum {
    RFW_MAT_HAS_DIFFUSE_MAP = 1u << 0, RFW_MAT_HAS_NORMAL_MAP = 1u << 1, RFW_MAT_HAS_ROUGHNESS_MAP = 1u << 2,
    RFW_MAT_HAS_METALLIC_MAP = 1u << 3, RFW_MAT_HAS_EMISSIVE_MAP = 1u << 4, RFW_MAT_HAS_SHEEN_MAP = 1u << 5
};

/* ---- borrowed-slice payloads (Rust `MeshData3D<'a>` etc. lowered to pointer + count) ---- */

/* crates/rfw-backend/src/structs.rs:332-341; C shape follows backends/metal/cpp/src/library.h:62-73 */
typedef struct {
    const rfw_vertex_3d* vertices;   uint32_t num_vertices;
    const rfw_rt_triangle* triangles; uint32_t num_triangles;
    const rfw_vertex_mesh* ranges;   uint32_t num_ranges;
    const rfw_joint_data* skin_data; uint32_t num_skin_data;
    uint32_t flags;
    rfw_aabb bounds;
} rfw_mesh_data_3d;

/* crates/rfw-backend/src/structs.rs:42-48; C shape follows backends/metal/cpp/src/library.h:80-89 */
typedef struct {
    rfw_aabb local_aabb;
    const rfw_mat4* matrices;  uint32_t num_matrices;
    const int32_t* skin_ids;   uint32_t num_skin_ids;
    const uint32_t* flags;     uint32_t num_flags;
} rfw_instances_data_3d;

/* crates/rfw-backend/src/structs.rs:69-77; backends/metal/cpp/src/library.h:109-116 */
typedef struct {
    uint32_t width;
    uint32_t height;
    uint32_t mip_levels;
    const uint8_t* bytes;     /* 4 bytes per texel, mips concatenated (structs.rs:79-121) */
    uint32_t format;
} rfw_texture_data;

/* crates/rfw-backend/src/structs.rs:6-11 */
typedef struct {
    const rfw_mat4* inverse_bind_matrices; uint32_t num_inverse_bind_matrices;
    const rfw_mat4* joint_matrices;        uint32_t num_joint_matrices;
} rfw_skin_data;

RFW_STATIC_ASSERT(sizeof(rfw_vec2) == 8 && sizeof(rfw_vec3) == 12 && sizeof(rfw_vec4) == 16, "glam vec sizes");
RFW_STATIC_ASSERT(sizeof(rfw_mat4) == 64, "Mat4");
RFW_STATIC_ASSERT(sizeof(rfw_aabb) == 32, "Aabb");
RFW_STATIC_ASSERT(sizeof(rfw_rt_triangle) == 176, "RTTriangle");
RFW_STATIC_ASSERT(offsetof(rfw_rt_triangle, normal) == 48 && offsetof(rfw_rt_triangle, n0) == 64, "RTTriangle normals");
RFW_STATIC_ASSERT(offsetof(rfw_rt_triangle, id) == 108 && offsetof(rfw_rt_triangle, tangent0) == 112, "RTTriangle tangents");
RFW_STATIC_ASSERT(offsetof(rfw_rt_triangle, light_id) == 160 && offsetof(rfw_rt_triangle, area) == 172, "RTTriangle tail");
RFW_STATIC_ASSERT(sizeof(rfw_vertex_3d) == 64 && offsetof(rfw_vertex_3d, tangent) == 48, "Vertex3D");
RFW_STATIC_ASSERT(sizeof(rfw_joint_data) == 32, "JointData");
RFW_STATIC_ASSERT(sizeof(rfw_vertex_mesh) == 48, "VertexMesh");
RFW_STATIC_ASSERT(sizeof(rfw_device_material) == 96 && offsetof(rfw_device_material, flags) == 64, "DeviceMaterial");
RFW_STATIC_ASSERT(sizeof(rfw_camera_view_3d) == 128 && offsetof(rfw_camera_view_3d, custom0) == 96, "CameraView3D");
RFW_STATIC_ASSERT(offsetof(rfw_camera_view_3d, lens_size) == 60 && offsetof(rfw_camera_view_3d, inv_width) == 72, "CameraView3D scalars");
RFW_STATIC_ASSERT(sizeof(rfw_area_light) == 96 && offsetof(rfw_area_light, radiance) == 64, "AreaLight");
RFW_STATIC_ASSERT(sizeof(rfw_point_light) == 32 && sizeof(rfw_directional_light) == 32, "Point/DirectionalLight");
RFW_STATIC_ASSERT(sizeof(rfw_spot_light) == 48 && offsetof(rfw_spot_light, energy) == 44, "SpotLight");

#endif /* RFW_POD_H */
