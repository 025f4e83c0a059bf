//! `impl Backend for HipBackend`: every trait method (crates/rfw-backend/src/lib.rs:35-82) forwards to the C entry point of the same name in
//! include/rfw_hip.h, the way backends/metal/src/lib.rs:58-268 does for the Metal backend.  Layout test as backends/metal/src/lib.rs:270-348.
// gpu-rt's blue-noise tables (Sobol sequence, scrambling and ranking tiles), compiled into this crate from the reference's own file:
// the sampler needs them for the first 256 samples of every image (backends/gpu-rt/shaders/ray_gen.comp:72-91,109-115)
#[path = "../../gpu-rt/src/blue_noise.rs"]
mod blue_noise;

use rfw::prelude::*;
use std::ffi::{c_void, CStr};
use std::os::raw::{c_char, c_double, c_int};

#[repr(C)]
#[derive(Default)]
pub struct Options { pub struct_size: u32, pub device: i32, pub max_path_length: u32, pub clamp_value: f32,
    pub rank: u32, pub world: u32, pub tile_size: u32, pub builder: u32, pub flags: u32, pub streams: u32, pub frames_in_flight: u32, pub max_batch: u32 }

#[repr(C)]
struct MeshData3DC { vertices: *const Vertex3D, num_vertices: u32, triangles: *const RTTriangle, num_triangles: u32,
    ranges: *const VertexMesh, num_ranges: u32, skin_data: *const JointData, num_skin_data: u32, flags: u32, bounds: Aabb }

#[repr(C)]
struct InstancesData3DC { local_aabb: Aabb, matrices: *const Mat4, num_matrices: u32, skin_ids: *const i32, num_skin_ids: u32,
    flags: *const u32, num_flags: u32 }

#[repr(C)]
struct TextureDataC { width: u32, height: u32, mip_levels: u32, bytes: *const u8, format: u32 }
#[repr(C)]
struct SkinDataC { inverse_bind_matrices: *const Mat4, num_inverse_bind_matrices: u32, joint_matrices: *const Mat4, num_joint_matrices: u32 }

extern "C" {
    fn rfw_hip_abi_version() -> u32;
    fn rfw_hip_create(width: u32, height: u32, scale: c_double, options: *const Options) -> *mut c_void;
    fn rfw_hip_destroy(instance: *mut c_void);
    fn rfw_hip_last_error(instance: *mut c_void) -> *const c_char;
    fn rfw_hip_set_3d_mesh(instance: *mut c_void, id: u32, data: *const MeshData3DC) -> c_int;
    fn rfw_hip_unload_3d_meshes(instance: *mut c_void, ids: *const u32, n: u32) -> c_int;
    fn rfw_hip_set_3d_instances(instance: *mut c_void, mesh: u32, data: *const InstancesData3DC) -> c_int;
    fn rfw_hip_set_materials(instance: *mut c_void, m: *const DeviceMaterial, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_set_textures(instance: *mut c_void, t: *const TextureDataC, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_synchronize(instance: *mut c_void) -> c_int;
    fn rfw_hip_render(instance: *mut c_void, view_2d: *const Mat4, view_3d: *const CameraView3D, mode: u32) -> c_int;
    fn rfw_hip_resize(instance: *mut c_void, w: u32, h: u32, scale: c_double) -> c_int;
    fn rfw_hip_set_point_lights(instance: *mut c_void, l: *const PointLight, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_set_spot_lights(instance: *mut c_void, l: *const SpotLight, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_set_area_lights(instance: *mut c_void, l: *const AreaLight, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_set_directional_lights(instance: *mut c_void, l: *const DirectionalLight, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_set_skybox(instance: *mut c_void, t: *const TextureDataC) -> c_int;
    fn rfw_hip_set_skins(instance: *mut c_void, s: *const SkinDataC, n: u32, changed: *const u32) -> c_int;
    fn rfw_hip_read_framebuffer(instance: *mut c_void, rgba: *mut f32, n_floats: u64) -> c_int;
    fn rfw_hip_set_blue_noise(instance: *mut c_void, table: *const u32, n_words: u32) -> c_int;
    // multi-GPU (one process per GPU): a communicator inside the library; render() then all-gathers the ranks' tiles itself (RCCL)
    fn rfw_hip_comm_unique_id(out128: *mut u8) -> c_int;
    fn rfw_hip_comm_init(instance: *mut c_void, id128: *const u8, rank: u32, world: u32) -> c_int;
    // presentation without stalling the frames in flight (what = 2: the Bgra8UnormSrgb image gpu-rt's swap chain would hold)
    fn rfw_hip_host_alloc(bytes: u64) -> *mut c_void;
    fn rfw_hip_host_free(ptr: *mut c_void);
    fn rfw_hip_download_frame(instance: *mut c_void, what: u32, frame: u32, host: *mut f32, n: u64) -> c_int;
    fn rfw_hip_wait_download(instance: *mut c_void, host: *const c_void) -> c_int;
}

pub struct HipBackend { instance: *mut c_void }
unsafe impl Send for HipBackend {}   // the library serialises calls internally and is thread-agnostic

impl HipBackend {
    fn check(&self, rc: c_int) {
        if rc != 0 { // trait methods return (): fail the way the other backends do
            let msg = unsafe { CStr::from_ptr(rfw_hip_last_error(self.instance)) }.to_string_lossy().into_owned();
            panic!("rfw-hip: {}", msg);
        }
    }
    /// BitSlice<Lsb0, usize> -> packed little-endian u32 words covering `n` elements (the library reads exactly n bits).  The trait does not
    /// promise `changed.len() == n` (rfw-scene builds the two from different ranges): bits past the slice read as CHANGED, so that a short
    /// slice can never make the library skip an element.
    fn words(changed: &BitSlice, n: usize) -> Vec<u32> {
        let mut w = vec![0u32; (n + 31) / 32];
        for i in 0..n { if changed.get(i).map(|b| *b).unwrap_or(true) { w[i / 32] |= 1 << (i % 32); } }
        w
    }
    /// One process per GPU: rank 0 calls `comm_unique_id()`, hands the 128 bytes to every rank (MPI, a file, a socket), every rank
    /// creates its backend with `Options { rank, world, .. }` and calls `comm_init`; from then on `render` leaves the complete frame on
    /// every rank (tiles traced locally, one RCCL all-gather per frame on the backend's own HIP stream).
    pub fn comm_unique_id() -> [u8; 128] { let mut id = [0u8; 128]; assert_eq!(unsafe { rfw_hip_comm_unique_id(id.as_mut_ptr()) }, 0); id }
    pub fn comm_init(&self, id: &[u8; 128], rank: u32, world: u32) { self.check(unsafe { rfw_hip_comm_init(self.instance, id.as_ptr(), rank, world) }) }
    /// Extension: the tonemapped frame (the trait itself has no read-back).
    pub fn read_framebuffer(&self, rgba: &mut [f32]) { self.check(unsafe { rfw_hip_read_framebuffer(self.instance, rgba.as_mut_ptr(), rgba.len() as u64) }) }
    /// Queue the copy of the frame just rendered, as B,G,R,A bytes, into a pinned buffer from `rfw_hip_host_alloc(width * height * 4)`;
    /// `wait_presented(buf)` before the bytes are handed to the window surface (or before the buffer is reused).
    pub fn present_into(&self, bgra: *mut u8, pixels: u64) { self.check(unsafe { rfw_hip_download_frame(self.instance, 2, 0, bgra as *mut f32, pixels) }) }
    pub fn wait_presented(&self, bgra: *const u8) { self.check(unsafe { rfw_hip_wait_download(self.instance, bgra as *const c_void) }) }
}

impl FromWindowHandle for HipBackend {
    fn init<W: HasRawWindowHandle>(_window: &W, width: u32, height: u32, scale: f64) -> Result<Box<Self>, Box<dyn std::error::Error>> {
        // the structs above mirror include/rfw_hip.h at RFW_HIP_ABI_VERSION 2: a library of another version is refused, not mis-read
        let abi = unsafe { rfw_hip_abi_version() };
        if abi != 2 { return Err(format!("librfw_hip.so speaks ABI version {}, this crate version 2", abi).into()); }
        let opts = Options { struct_size: std::mem::size_of::<Options>() as u32, device: -1, ..Default::default() };
        let instance = unsafe { rfw_hip_create(width, height, scale, &opts) };
        if instance.is_null() {
            let msg = unsafe { CStr::from_ptr(rfw_hip_last_error(std::ptr::null_mut())) }.to_string_lossy().into_owned();
            return Err(msg.into());
        }
        let backend = Self { instance };
        let tables = blue_noise::create_blue_noise_buffer();   // 5 x 65536 words (backends/gpu-rt/src/blue_noise.rs:40970-41005)
        backend.check(unsafe { rfw_hip_set_blue_noise(instance, tables.as_ptr(), tables.len() as u32) });
        Ok(Box::new(backend))
    }
}

impl Backend for HipBackend {
    fn set_2d_mesh(&mut self, _id: usize, _data: MeshData2D<'_>) {}
    fn set_2d_instances(&mut self, _mesh: usize, _instances: InstancesData2D<'_>) {}

    fn set_3d_mesh(&mut self, id: usize, data: MeshData3D<'_>) {
        let d = MeshData3DC { vertices: data.vertices.as_ptr(), num_vertices: data.vertices.len() as u32,
            triangles: data.triangles.as_ptr(), num_triangles: data.triangles.len() as u32,
            ranges: data.ranges.as_ptr(), num_ranges: data.ranges.len() as u32,
            skin_data: data.skin_data.as_ptr(), num_skin_data: data.skin_data.len() as u32,
            flags: data.flags.bits(), bounds: data.bounds };
        self.check(unsafe { rfw_hip_set_3d_mesh(self.instance, id as u32, &d) });
    }
    fn unload_3d_meshes(&mut self, ids: &[usize]) {
        let ids: Vec<u32> = ids.iter().map(|i| *i as u32).collect();
        self.check(unsafe { rfw_hip_unload_3d_meshes(self.instance, ids.as_ptr(), ids.len() as u32) });
    }
    fn set_3d_instances(&mut self, mesh: usize, instances: InstancesData3D<'_>) {
        let d = InstancesData3DC { local_aabb: instances.local_aabb,
            matrices: instances.matrices.as_ptr(), num_matrices: instances.matrices.len() as u32,
            skin_ids: instances.skin_ids.as_ptr() as *const i32, num_skin_ids: instances.skin_ids.len() as u32,   // SkinID(i32), repr(transparent)
            flags: instances.flags.as_ptr() as *const u32, num_flags: instances.flags.len() as u32 };              // InstanceFlags3D repr(transparent) u32
        self.check(unsafe { rfw_hip_set_3d_instances(self.instance, mesh as u32, &d) });
    }
    fn set_materials(&mut self, materials: &[DeviceMaterial], changed: &BitSlice) {
        let w = Self::words(changed, materials.len());
        self.check(unsafe { rfw_hip_set_materials(self.instance, materials.as_ptr(), materials.len() as u32, w.as_ptr()) });
    }
    fn set_textures(&mut self, textures: &[TextureData<'_>], changed: &BitSlice) {
        let t: Vec<TextureDataC> = textures.iter().map(|t| TextureDataC { width: t.width, height: t.height, mip_levels: t.mip_levels,
            bytes: t.bytes.as_ptr(), format: t.format as u32 }).collect();
        let w = Self::words(changed, t.len());
        self.check(unsafe { rfw_hip_set_textures(self.instance, t.as_ptr(), t.len() as u32, w.as_ptr()) });
    }
    fn synchronize(&mut self) { self.check(unsafe { rfw_hip_synchronize(self.instance) }); }
    fn render(&mut self, view_2d: CameraView2D, view_3d: CameraView3D, mode: RenderMode) {
        self.check(unsafe { rfw_hip_render(self.instance, &view_2d.matrix, &view_3d, mode as u32) });
    }
    fn resize(&mut self, window_size: (u32, u32), scale_factor: f64) {
        self.check(unsafe { rfw_hip_resize(self.instance, window_size.0, window_size.1, scale_factor) });
    }
    fn set_point_lights(&mut self, lights: &[PointLight], changed: &BitSlice) {
        let w = Self::words(changed, lights.len());
        self.check(unsafe { rfw_hip_set_point_lights(self.instance, lights.as_ptr(), lights.len() as u32, w.as_ptr()) });
    }
    fn set_spot_lights(&mut self, lights: &[SpotLight], changed: &BitSlice) {
        let w = Self::words(changed, lights.len());
        self.check(unsafe { rfw_hip_set_spot_lights(self.instance, lights.as_ptr(), lights.len() as u32, w.as_ptr()) });
    }
    fn set_area_lights(&mut self, lights: &[AreaLight], changed: &BitSlice) {
        let w = Self::words(changed, lights.len());
        self.check(unsafe { rfw_hip_set_area_lights(self.instance, lights.as_ptr(), lights.len() as u32, w.as_ptr()) });
    }
    fn set_directional_lights(&mut self, lights: &[DirectionalLight], changed: &BitSlice) {
        let w = Self::words(changed, lights.len());
        self.check(unsafe { rfw_hip_set_directional_lights(self.instance, lights.as_ptr(), lights.len() as u32, w.as_ptr()) });
    }
    fn set_skybox(&mut self, skybox: TextureData<'_>) {
        let t = TextureDataC { width: skybox.width, height: skybox.height, mip_levels: skybox.mip_levels, bytes: skybox.bytes.as_ptr(), format: skybox.format as u32 };
        self.check(unsafe { rfw_hip_set_skybox(self.instance, &t) });
    }
    fn set_skins(&mut self, skins: &[SkinData<'_>], changed: &BitSlice) {
        let c: Vec<SkinDataC> = skins.iter().map(|s| SkinDataC {
            inverse_bind_matrices: s.inverse_bind_matrices.as_ptr(), num_inverse_bind_matrices: s.inverse_bind_matrices.len() as u32,
            joint_matrices: s.joint_matrices.as_ptr(), num_joint_matrices: s.joint_matrices.len() as u32 }).collect();
        let w = Self::words(changed, c.len());
        self.check(unsafe { rfw_hip_set_skins(self.instance, c.as_ptr(), c.len() as u32, w.as_ptr()) });
    }
}

impl Drop for HipBackend { fn drop(&mut self) { unsafe { rfw_hip_destroy(self.instance) } } }

#[cfg(test)]
mod tests {
    use rfw::prelude::*;
    // the C side asserts the same numbers (include/rfw_pod.h); cf. backends/metal/src/lib.rs:270-348
    #[test]
    fn test_layout() {
        assert_eq!(std::mem::size_of::<RTTriangle>(), 176);
        assert_eq!(std::mem::size_of::<Vertex3D>(), 64);
        assert_eq!(std::mem::size_of::<CameraView3D>(), 128);
        assert_eq!(std::mem::size_of::<DeviceMaterial>(), 96);
        assert_eq!(std::mem::size_of::<AreaLight>(), 96);
        assert_eq!(std::mem::size_of::<PointLight>(), 32);
        assert_eq!(std::mem::size_of::<SpotLight>(), 48);
        assert_eq!(std::mem::size_of::<DirectionalLight>(), 32);
        assert_eq!(std::mem::size_of::<Aabb>(), 32);
        assert_eq!(std::mem::size_of::<VertexMesh>(), 48);
        assert_eq!(std::mem::size_of::<JointData>(), 32);
        assert_eq!(std::mem::size_of::<Mat4>(), 64);
    }
}
