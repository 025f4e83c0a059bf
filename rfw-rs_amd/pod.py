"""ctypes mirrors of include/rfw_pod.h and include/rfw_hip.h (sizes asserted at import)."""
import ctypes as C

f32, u32, i32, u64, u8 = C.c_float, C.c_uint32, C.c_int32, C.c_uint64, C.c_uint8


class Vec2(C.Structure):
    _fields_ = [("x", f32), ("y", f32)]


class Vec3(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32)]

    def tolist(self):
        return [self.x, self.y, self.z]


class Vec4(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32), ("w", f32)]


class Mat4(C.Structure):
    _fields_ = [("m", f32 * 16)]


class Aabb(C.Structure):
    _fields_ = [("min", f32 * 3), ("extra1", i32), ("max", f32 * 3), ("extra2", i32)]


class RTTriangle(C.Structure):
    _fields_ = [
        ("vertex0", Vec3), ("u0", f32), ("vertex1", Vec3), ("u1", f32), ("vertex2", Vec3), ("u2", f32),
        ("normal", Vec3), ("v0", f32), ("n0", Vec3), ("v1", f32), ("n1", Vec3), ("v2", f32),
        ("n2", Vec3), ("id", i32), ("tangent0", Vec4), ("tangent1", Vec4), ("tangent2", Vec4),
        ("light_id", i32), ("mat_id", i32), ("lod", f32), ("area", f32),
    ]


class Vertex3D(C.Structure):
    _fields_ = [("vertex", Vec4), ("normal", Vec3), ("mat_id", u32), ("uv", Vec2), ("pad0", f32), ("pad1", f32), ("tangent", Vec4)]


class JointData(C.Structure):
    _fields_ = [("joint", u32 * 4), ("weight", Vec4)]


class VertexMesh(C.Structure):
    _fields_ = [("bounds", Aabb), ("first", u32), ("last", u32), ("mat_id", u32), ("padding", u32)]


class DeviceMaterial(C.Structure):
    _fields_ = [
        ("color", f32 * 4), ("absorption", f32 * 4), ("specular", f32 * 4), ("parameters", u32 * 4), ("flags", u32),
        ("diffuse_map", i32), ("normal_map", i32), ("metallic_roughness_map", i32), ("emissive_map", i32), ("sheen_map", i32),
        ("_dummy", i32 * 2),
    ]


class CameraView3D(C.Structure):
    _fields_ = [
        ("pos", Vec3), ("right", Vec3), ("up", Vec3), ("p1", Vec3), ("direction", Vec3), ("lens_size", f32),
        ("spread_angle", f32), ("epsilon", f32), ("inv_width", f32), ("inv_height", f32), ("near_plane", f32),
        ("far_plane", f32), ("aspect_ratio", f32), ("fov", f32), ("custom0", Vec4), ("custom1", Vec4),
    ]


class AreaLight(C.Structure):
    _fields_ = [
        ("position", Vec3), ("energy", f32), ("normal", Vec3), ("area", f32), ("vertex0", Vec3), ("inst_idx", i32),
        ("vertex1", Vec3), ("mesh_id", i32), ("radiance", Vec3), ("_dummy1", i32), ("vertex2", Vec3), ("_dummy2", i32),
    ]


class PointLight(C.Structure):
    _fields_ = [("position", Vec3), ("energy", f32), ("radiance", Vec3), ("_dummy", f32)]


class SpotLight(C.Structure):
    _fields_ = [("position", Vec3), ("cos_inner", f32), ("radiance", Vec3), ("cos_outer", f32), ("direction", Vec3), ("energy", f32)]


class DirectionalLight(C.Structure):
    _fields_ = [("direction", Vec3), ("energy", f32), ("radiance", Vec3), ("_dummy", f32)]


class MeshData3D(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(Vertex3D)), ("num_vertices", u32), ("triangles", C.POINTER(RTTriangle)), ("num_triangles", u32),
        ("ranges", C.POINTER(VertexMesh)), ("num_ranges", u32), ("skin_data", C.POINTER(JointData)), ("num_skin_data", u32),
        ("flags", u32), ("bounds", Aabb),
    ]


class InstancesData3D(C.Structure):
    _fields_ = [
        ("local_aabb", Aabb), ("matrices", C.POINTER(Mat4)), ("num_matrices", u32), ("skin_ids", C.POINTER(i32)), ("num_skin_ids", u32),
        ("flags", C.POINTER(u32)), ("num_flags", u32),
    ]


class TextureData(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("mip_levels", u32), ("bytes", C.POINTER(u8)), ("format", u32)]


class SkinData(C.Structure):
    _fields_ = [("inverse_bind_matrices", C.POINTER(Mat4)), ("num_inverse_bind_matrices", u32), ("joint_matrices", C.POINTER(Mat4)), ("num_joint_matrices", u32)]


class HipOptions(C.Structure):
    _fields_ = [
        ("struct_size", u32), ("device", i32), ("max_path_length", u32), ("clamp_value", f32), ("rank", u32), ("world", u32),
        ("tile_size", u32), ("builder", u32), ("flags", u32), ("streams", u32), ("frames_in_flight", u32), ("max_batch", u32),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("primary_rays", u64), ("extension_rays", u64), ("shadow_rays", u64), ("nodes_visited", u64 * 3), ("tris_tested", u64 * 3),
        ("instances_entered", u64 * 3), ("ms_total", f32), ("ms_trace_primary", f32), ("ms_trace_extend", f32), ("ms_trace_shadow", f32),
        ("ms_shade", f32), ("ms_other", f32), ("sample_count", u32), ("bounces", u32), ("substreams", u32), ("pad", u32),
        ("node_test_executions", u64 * 3), ("tri_test_executions", u64 * 3), ("wave_max_nodes", u64 * 3),
        ("uniform_node_test_executions", u64 * 3),
    ]


class SceneStats(C.Structure):
    _fields_ = [
        ("triangles", u64), ("instances", u64), ("blas_nodes", u64), ("tlas_nodes", u64), ("node_bytes", u32), ("tri_bytes", u32),
        ("ms_blas_build", f32), ("ms_tlas_build", f32), ("ms_blas_upload", f32), ("ms_blas_kernels", f32),
        ("blas_upload_bytes", u64), ("blas_kernel_bytes", u64), ("split_references", u64),
        ("accel_bytes", u64), ("packet_copies", u32), ("pad", u32),
    ]


class Hit(C.Structure):
    _fields_ = [("inst", i32), ("tri", i32), ("t", f32), ("u", f32), ("v", f32)]


# the reference's only boundary test, restated: backends/metal/src/lib.rs:270-348 (size_of Rust == size_of C)
EXPECTED_SIZES = {
    Vec2: 8, Vec3: 12, Vec4: 16, Mat4: 64, Aabb: 32, RTTriangle: 176, Vertex3D: 64, JointData: 32, VertexMesh: 48,
    DeviceMaterial: 96, CameraView3D: 128, AreaLight: 96, PointLight: 32, SpotLight: 48, DirectionalLight: 32,
}
for _t, _n in EXPECTED_SIZES.items():
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)

RFW_HIP_FLAG_NO_NEE = 1
RFW_HIP_FLAG_COUNT_TRAVERSAL = 2
RFW_HIP_BUILDER_AUTO = 0
RFW_HIP_BUILDER_HOST_SAH = 1
RFW_HIP_BUILDER_DEVICE_LBVH = 2
RFW_HIP_BUILDER_DEVICE_SAH = 3
