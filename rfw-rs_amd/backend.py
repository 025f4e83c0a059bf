"""Python mirror of the reference's plugin interface for the path — `rfw_backend::Backend`
(crates/rfw-backend/src/lib.rs:35-82) — over the C ABI of include/rfw_hip.h.  Method names,
argument meaning and order follow the trait; every method is one call into librfw_hip.so
(hand-written HIP for gfx950).  There is no Python or CPU fallback: if the library is missing
or no HIP device is present, construction raises."""
import ctypes as C
import os

import numpy as np

from . import pod
from .scene import BackendTable

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("RFW_HIP_LIB") or os.path.join(_HERE, "csrc", "librfw_hip.so")  # RFW_HIP_LIB: experiment builds (make -C rfw-rs_amd/csrc variant VARIANT=x VFLAGS=-D...)

EXPORTS = [
    "rfw_hip_create", "rfw_hip_destroy", "rfw_hip_last_error", "rfw_hip_abi_version", "rfw_hip_selftest_bvh", "rfw_hip_selftest_splits", "rfw_hip_selftest_index_magic",
    "rfw_hip_set_2d_mesh", "rfw_hip_set_2d_instances", "rfw_hip_set_3d_mesh", "rfw_hip_unload_3d_meshes",
    "rfw_hip_set_3d_instances", "rfw_hip_set_materials", "rfw_hip_set_textures", "rfw_hip_synchronize",
    "rfw_hip_render", "rfw_hip_resize", "rfw_hip_set_point_lights", "rfw_hip_set_spot_lights",
    "rfw_hip_set_area_lights", "rfw_hip_set_directional_lights", "rfw_hip_set_skybox", "rfw_hip_set_skins",
    "rfw_hip_reset_accumulation", "rfw_hip_set_option", "rfw_hip_read_framebuffer", "rfw_hip_read_accumulator",
    "rfw_hip_get_frame_stats", "rfw_hip_drain_timing", "rfw_hip_get_scene_stats", "rfw_hip_set_stream", "rfw_hip_get_stream", "rfw_hip_device_synchronize",
    "rfw_hip_shard_info", "rfw_hip_set_slab_output", "rfw_hip_assemble_frame", "rfw_hip_intersect", "rfw_hip_occludes", "rfw_hip_debug_occludes_depth",
    "rfw_hip_debug_read", "rfw_hip_bandwidth_probe", "rfw_hip_depth_test", "rfw_hip_render_batch", "rfw_hip_assemble_batch",
    "rfw_hip_read_framebuffer_at", "rfw_hip_read_accumulator_at", "rfw_hip_host_alloc", "rfw_hip_host_free", "rfw_hip_download_frame",
    "rfw_hip_wait_downloads", "rfw_hip_wait_download", "rfw_hip_srgb_steps", "rfw_hip_render_samples", "rfw_hip_set_blue_noise", "rfw_hip_debug_eval_shading", "rfw_hip_comm_unique_id", "rfw_hip_comm_init", "rfw_hip_comm_init_loopback", "rfw_hip_comm_destroy", "rfw_hip_p2p_export", "rfw_hip_p2p_connect", "rfw_hip_p2p_disconnect", "rfw_hip_intersect4", "rfw_hip_occludes4", "rfw_hip_debug_lbvh_stress", "rfw_hip_issue_probe",
]

_lib = None


ABI_VERSION = 2  # RFW_HIP_ABI_VERSION of include/rfw_hip.h this file's structs (pod.py) were written against


def hip_lib():
    """Load librfw_hip.so (loudly: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"HIP extension {HIP_LIB} is missing — build it with __graft_entry__.build(); there is no CPU fallback")
        try:
            # torch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's): whichever is loaded first serves the
            # whole process.  Load torch's first so that torch (device memory, streams, torch.distributed) and this
            # library share ONE HIP runtime and stream handles can cross between them.
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(HIP_LIB)
        vp, u32, u64, f32, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_char_p
        l.rfw_hip_create.restype = vp
        l.rfw_hip_create.argtypes = [u32, u32, C.c_double, C.POINTER(pod.HipOptions)]
        l.rfw_hip_destroy.argtypes = [vp]
        l.rfw_hip_destroy.restype = None
        l.rfw_hip_last_error.argtypes = [vp]
        l.rfw_hip_last_error.restype = cp
        l.rfw_hip_abi_version.restype = u32
        if l.rfw_hip_abi_version() != ABI_VERSION:  # (pod.py mirrors the structs of ONE version of include/rfw_hip.h)
            raise RuntimeError(f"{HIP_LIB} speaks ABI version {l.rfw_hip_abi_version()}, this binding version {ABI_VERSION}: rebuild the library")
        l.rfw_hip_selftest_bvh.argtypes = [vp, u32, u32, u32, C.POINTER(u32)]
        l.rfw_hip_selftest_bvh.restype = C.c_int64
        l.rfw_hip_selftest_splits.argtypes = [vp, u32, C.c_float, u32, vp, u32, vp, u32, C.POINTER(u32)]
        l.rfw_hip_selftest_splits.restype = C.c_int64
        l.rfw_hip_selftest_index_magic.argtypes = [u32, C.c_uint64]
        l.rfw_hip_selftest_index_magic.restype = u32
        l.rfw_hip_set_2d_mesh.argtypes = [vp, u32, vp, u32, C.c_int32]
        l.rfw_hip_set_2d_instances.argtypes = [vp, u32, vp, u32]
        l.rfw_hip_set_3d_mesh.argtypes = [vp, u32, C.POINTER(pod.MeshData3D)]
        l.rfw_hip_unload_3d_meshes.argtypes = [vp, C.POINTER(u32), u32]
        l.rfw_hip_set_3d_instances.argtypes = [vp, u32, C.POINTER(pod.InstancesData3D)]
        l.rfw_hip_set_materials.argtypes = [vp, vp, u32, vp]
        l.rfw_hip_set_textures.argtypes = [vp, vp, u32, vp]
        l.rfw_hip_synchronize.argtypes = [vp]
        l.rfw_hip_render.argtypes = [vp, C.POINTER(pod.Mat4), C.POINTER(pod.CameraView3D), u32]
        l.rfw_hip_resize.argtypes = [vp, u32, u32, C.c_double]
        for n in ("point", "spot", "area", "directional"):
            getattr(l, f"rfw_hip_set_{n}_lights").argtypes = [vp, vp, u32, vp]
        l.rfw_hip_set_skybox.argtypes = [vp, C.POINTER(pod.TextureData)]
        l.rfw_hip_set_skins.argtypes = [vp, vp, u32, vp]
        l.rfw_hip_reset_accumulation.argtypes = [vp]
        l.rfw_hip_set_option.argtypes = [vp, cp, C.c_double]
        l.rfw_hip_read_framebuffer.argtypes = [vp, vp, u64]
        l.rfw_hip_read_accumulator.argtypes = [vp, vp, u64]
        l.rfw_hip_get_frame_stats.argtypes = [vp, C.POINTER(pod.FrameStats)]
        l.rfw_hip_get_scene_stats.argtypes = [vp, C.POINTER(pod.SceneStats)]
        l.rfw_hip_drain_timing.argtypes = [vp, C.POINTER(pod.FrameStats), C.POINTER(u32)]
        l.rfw_hip_set_stream.argtypes = [vp, vp]
        l.rfw_hip_device_synchronize.argtypes = [vp]
        l.rfw_hip_get_stream.argtypes = [vp]
        l.rfw_hip_get_stream.restype = vp
        l.rfw_hip_shard_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]
        l.rfw_hip_set_slab_output.argtypes = [vp, vp]
        l.rfw_hip_assemble_frame.argtypes = [vp, vp]
        l.rfw_hip_intersect.argtypes = [vp, vp, vp, f32, f32, u64, vp]
        l.rfw_hip_depth_test.argtypes = [vp, vp, vp, f32, f32, u64, vp, vp]
        l.rfw_hip_render_batch.argtypes = [vp, vp, C.c_uint32]
        l.rfw_hip_assemble_batch.argtypes = [vp, vp, C.c_uint32]
        l.rfw_hip_render_samples.argtypes = [vp, C.POINTER(pod.CameraView3D), C.c_uint32]
        l.rfw_hip_set_blue_noise.argtypes = [vp, vp, C.c_uint32]
        l.rfw_hip_debug_eval_shading.argtypes = [vp, C.c_int, u64, vp, vp]
        l.rfw_hip_comm_unique_id.argtypes = [vp]
        l.rfw_hip_comm_init.argtypes = [vp, vp, u32, u32]
        l.rfw_hip_comm_destroy.argtypes = [vp]
        l.rfw_hip_comm_init_loopback.argtypes = [vp, C.c_uint64, u32, u32]
        l.rfw_hip_p2p_export.argtypes = [vp, vp]
        l.rfw_hip_p2p_connect.argtypes = [vp, vp]
        l.rfw_hip_p2p_disconnect.argtypes = [vp]
        l.rfw_hip_intersect4.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        l.rfw_hip_occludes4.argtypes = [vp, vp, vp, vp, vp, vp]
        l.rfw_hip_read_framebuffer_at.argtypes = [vp, C.c_uint32, vp, u64]
        l.rfw_hip_read_accumulator_at.argtypes = [vp, C.c_uint32, vp, u64]
        l.rfw_hip_host_alloc.restype = vp
        l.rfw_hip_host_alloc.argtypes = [u64]
        l.rfw_hip_host_free.restype = None
        l.rfw_hip_host_free.argtypes = [vp]
        l.rfw_hip_download_frame.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u64]
        l.rfw_hip_wait_downloads.argtypes = [vp]
        l.rfw_hip_wait_download.argtypes = [vp, vp]
        l.rfw_hip_srgb_steps.restype = None
        l.rfw_hip_srgb_steps.argtypes = [vp]
        l.rfw_hip_occludes.argtypes = [vp, vp, vp, f32, vp, u64, vp]
        l.rfw_hip_debug_occludes_depth.argtypes = [vp, vp, vp, f32, vp, u64, vp, vp]
        l.rfw_hip_debug_read.argtypes = [vp, cp, vp, u64, C.POINTER(u64)]
        l.rfw_hip_debug_lbvh_stress.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
        l.rfw_hip_bandwidth_probe.argtypes = [vp, u64, C.c_uint32, C.POINTER(C.c_double)]
        l.rfw_hip_issue_probe.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
        _lib = l
    return _lib


HIT_DTYPE = np.dtype([("inst", "<i4"), ("tri", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


class BackendError(RuntimeError):
    pass


class HipBackend:
    """`impl Backend for HipBackend` — see crates/rfw-backend/src/lib.rs:35-82 for each method."""

    @classmethod
    def init(cls, width, height, scale=1.0, **options):
        """FromWindowHandle::init(window, width, height, scale) — headless, the window handle is dropped."""
        return cls(width, height, scale, **options)

    def __init__(self, width, height, scale=1.0, device=-1, max_path_length=0, clamp_value=0.0, rank=0, world=1,
                 tile_size=0, builder=pod.RFW_HIP_BUILDER_AUTO, flags=0, streams=0, frames_in_flight=0, max_batch=0):
        self._l = hip_lib()
        o = pod.HipOptions(C.sizeof(pod.HipOptions), device, max_path_length, clamp_value, rank, world, tile_size, builder, flags, streams, frames_in_flight, max_batch)
        h = self._l.rfw_hip_create(width, height, scale, C.byref(o))
        if not h:
            raise BackendError("rfw_hip_create failed: " + self._l.rfw_hip_last_error(None).decode())
        self._h = C.c_void_p(h)
        self._pinned = {}
        self.width, self.height = width, height
        self.rank, self.world = rank, max(world, 1)

    def close(self):
        if getattr(self, "_h", None):
            self._l.rfw_hip_destroy(self._h)  # waits for queued work (incl. downloads) first
            for p in self._pinned.values():
                self._l.rfw_hip_host_free(p)
            self._pinned = {}
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self._l.rfw_hip_last_error(self._h).decode()

    def _check(self, rc):
        if rc != 0:
            raise BackendError(f"rfw_hip error {rc}: {self.last_error()}")

    def table(self):
        t = BackendTable()
        t.instance = self._h
        for name in ("set_3d_mesh", "unload_3d_meshes", "set_3d_instances", "set_materials", "synchronize",
                     "set_point_lights", "set_spot_lights", "set_area_lights", "set_directional_lights", "set_textures", "set_skybox", "set_skins"):
            setattr(t, name, C.cast(getattr(self._l, "rfw_hip_" + name), C.c_void_p))
        return t

    # ---- trait methods (same order as the trait) ----
    def set_2d_mesh(self, id, data=None):
        self._check(self._l.rfw_hip_set_2d_mesh(self._h, id, None, 0, -1))

    def set_2d_instances(self, mesh, instances=None):
        self._check(self._l.rfw_hip_set_2d_instances(self._h, mesh, None, 0))

    def set_3d_mesh(self, id, data):
        self._check(self._l.rfw_hip_set_3d_mesh(self._h, id, C.byref(data)))

    def unload_3d_meshes(self, ids):
        arr = (C.c_uint32 * len(ids))(*ids)
        self._check(self._l.rfw_hip_unload_3d_meshes(self._h, arr, len(ids)))

    def set_3d_instances(self, mesh, instances):
        self._check(self._l.rfw_hip_set_3d_instances(self._h, mesh, C.byref(instances)))

    def set_materials(self, materials, changed=None):
        arr = (pod.DeviceMaterial * len(materials))(*materials)
        self._check(self._l.rfw_hip_set_materials(self._h, arr, len(materials), None))

    def set_textures(self, textures, changed=None):
        """changed: optional list of indices whose bit is set in the trait's `changed` BitSlice (None: everything)."""
        arr = (pod.TextureData * len(textures))(*textures)
        bits = None
        if changed is not None:
            bits = (C.c_uint32 * ((len(textures) + 31) // 32 or 1))()
            for k in changed:
                bits[k // 32] |= 1 << (k % 32)
        self._check(self._l.rfw_hip_set_textures(self._h, arr, len(textures), bits))

    def synchronize(self):
        self._check(self._l.rfw_hip_synchronize(self._h))

    def render(self, view_3d, view_2d=None, mode=0):
        self._check(self._l.rfw_hip_render(self._h, None, C.byref(view_3d), mode))

    def resize(self, window_size, scale_factor=1.0):
        self._check(self._l.rfw_hip_resize(self._h, window_size[0], window_size[1], scale_factor))
        self.width, self.height = window_size

    def _set_lights(self, kind, ctype, lights):
        arr = (ctype * len(lights))(*lights)
        self._check(getattr(self._l, f"rfw_hip_set_{kind}_lights")(self._h, arr, len(lights), None))

    def set_point_lights(self, lights, changed=None):
        self._set_lights("point", pod.PointLight, lights)

    def set_spot_lights(self, lights, changed=None):
        self._set_lights("spot", pod.SpotLight, lights)

    def set_area_lights(self, lights, changed=None):
        self._set_lights("area", pod.AreaLight, lights)

    def set_directional_lights(self, lights, changed=None):
        self._set_lights("directional", pod.DirectionalLight, lights)

    def set_skybox(self, skybox):
        self._check(self._l.rfw_hip_set_skybox(self._h, C.byref(skybox)))

    def set_skins(self, skins, changed=None):
        self._check(self._l.rfw_hip_set_skins(self._h, None, 0, None))

    # ---- extensions ----
    def reset_accumulation(self):
        self._check(self._l.rfw_hip_reset_accumulation(self._h))

    def set_option(self, key, value):
        self._check(self._l.rfw_hip_set_option(self._h, key.encode(), float(value)))

    def render_batch(self, views):
        """k independent new images, one per view, in one launch per stage (options.max_batch >= k)."""
        arr = (pod.CameraView3D * len(views))(*views)
        self._check(self._l.rfw_hip_render_batch(self._h, arr, len(views)))

    def render_samples(self, view, count):
        """`count` consecutive samples of the image of `view` in one launch per stage (options.max_batch >= count)."""
        self._check(self._l.rfw_hip_render_samples(self._h, C.byref(view), count))

    def set_blue_noise(self, table):
        """The blue-noise sampler's tables: the 5 * 65536 words of gpu_rt::blue_noise::create_blue_noise_buffer(), or None to clear."""
        if table is None:
            self._check(self._l.rfw_hip_set_blue_noise(self._h, None, 0))
        else:
            t = np.ascontiguousarray(table, dtype=np.uint32)
            self._check(self._l.rfw_hip_set_blue_noise(self._h, t.ctypes.data, t.size))

    def eval_shading(self, op, inputs):
        """Test-only: the shade kernel's device functions on caller-supplied inputs, (n, 48) float32 -> (n, 12) float32."""
        a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 48)
        out = np.zeros((len(a), 12), dtype=np.float32)
        self._check(self._l.rfw_hip_debug_eval_shading(self._h, op, len(a), a.ctypes.data, out.ctypes.data))
        return out

    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands the bytes to every rank)."""
        buf = (C.c_uint8 * 128)()
        if hip_lib().rfw_hip_comm_unique_id(buf) != 0:
            raise BackendError("rfw_hip_comm_unique_id failed: " + hip_lib().rfw_hip_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        """Collective: the instance gets its own RCCL communicator; render() then gathers and assembles the frame itself."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._l.rfw_hip_comm_init(self._h, buf, rank, world))

    def comm_init_loopback(self, hub_key, rank, world):
        """TEST TRANSPORT: the instances of one process (one per rank, one device) exchange their tiles through hub `hub_key` instead of a
        communicator — everything around the collective runs as with comm_init (include/rfw_hip.h)."""
        self._check(self._l.rfw_hip_comm_init_loopback(self._h, hub_key, rank, world))

    def comm_destroy(self):
        self._check(self._l.rfw_hip_comm_destroy(self._h))

    P2P_HANDLE_BYTES = 256

    def p2p_export(self):
        """This rank's 256-byte handle for the exchange by peer stores (every rank hands its handle to every rank)."""
        buf = (C.c_uint8 * self.P2P_HANDLE_BYTES)()
        self._check(self._l.rfw_hip_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_connect(self, handles):
        """handles: every rank's handle in rank order.  render() then stores this rank's tiles straight into the destinations' buffers."""
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._check(self._l.rfw_hip_p2p_connect(self._h, buf))

    def p2p_disconnect(self):
        self._check(self._l.rfw_hip_p2p_disconnect(self._h))

    def assemble_batch(self, gathered_ptr, count):
        self._check(self._l.rfw_hip_assemble_batch(self._h, C.c_void_p(gathered_ptr), count))

    def accumulator_at(self, frame):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._l.rfw_hip_read_accumulator_at(self._h, frame, a.ctypes.data, a.size))
        return a

    def framebuffer_at(self, frame):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._l.rfw_hip_read_framebuffer_at(self._h, frame, a.ctypes.data, a.size))
        return a

    def host_frame(self, presented=False):
        """A pinned (h, w, 4) array for download_frame: float32, or uint8 B,G,R,A for the presented frame; freed with free_host_frame."""
        n = self.height * self.width * (1 if presented else 4)
        p = self._l.rfw_hip_host_alloc(n * 4)
        if not p:
            raise BackendError("rfw_hip_host_alloc failed")
        if presented:
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.height, self.width, 4))
        else:
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.height, self.width, 4))
        self._pinned[a.ctypes.data] = p
        return a

    def free_host_frame(self, a):
        self._l.rfw_hip_host_free(self._pinned.pop(a.ctypes.data))

    def download_frame(self, dst, accumulator=False, frame=0):
        """Queue the copy of the latest frame into `dst` (from host_frame) behind its kernels; valid after wait_downloads()."""
        if dst.dtype == np.uint8:  # the presented (swap-chain) frame
            self._check(self._l.rfw_hip_download_frame(self._h, 2, frame, dst.ctypes.data, dst.size // 4))
        else:
            self._check(self._l.rfw_hip_download_frame(self._h, 1 if accumulator else 0, frame, dst.ctypes.data, dst.size))

    def srgb_steps(self):
        t = np.empty(255, np.float32)
        self._l.rfw_hip_srgb_steps(t.ctypes.data)
        return t

    def wait_downloads(self, dst=None):
        """Wait for every queued download, or only for the one into `dst`."""
        if dst is None:
            self._check(self._l.rfw_hip_wait_downloads(self._h))
        else:
            self._check(self._l.rfw_hip_wait_download(self._h, dst.ctypes.data))

    def framebuffer(self):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._l.rfw_hip_read_framebuffer(self._h, a.ctypes.data, a.size))
        return a

    def accumulator(self):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._l.rfw_hip_read_accumulator(self._h, a.ctypes.data, a.size))
        return a

    def frame_stats(self):
        s = pod.FrameStats()
        self._check(self._l.rfw_hip_get_frame_stats(self._h, C.byref(s)))
        return {n: (list(getattr(s, n)) if n in ("nodes_visited", "tris_tested", "instances_entered", "node_test_executions", "tri_test_executions", "wave_max_nodes", "uniform_node_test_executions") else getattr(s, n)) for n, _ in pod.FrameStats._fields_}

    def drain_timing(self):
        """Summed per-kernel HIP-event milliseconds of the frames rendered since the last drain, and their count."""
        s, n = pod.FrameStats(), C.c_uint32(0)
        self._check(self._l.rfw_hip_drain_timing(self._h, C.byref(s), C.byref(n)))
        return {k: getattr(s, k) for k, _ in pod.FrameStats._fields_ if k.startswith("ms_")}, int(n.value)

    def scene_stats(self):
        s = pod.SceneStats()
        self._check(self._l.rfw_hip_get_scene_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in pod.SceneStats._fields_}

    def set_stream(self, stream_handle):
        self._check(self._l.rfw_hip_set_stream(self._h, C.c_void_p(stream_handle)))

    def stream_handle(self):
        """hipStream_t of this instance as an integer (wrap with torch.cuda.ExternalStream to order torch work against it)."""
        return int(self._l.rfw_hip_get_stream(self._h) or 0)

    def device_synchronize(self):
        self._check(self._l.rfw_hip_device_synchronize(self._h))

    def shard_info(self):
        f, a, b = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._l.rfw_hip_shard_info(self._h, C.byref(f), C.byref(a), C.byref(b)))
        return {"slab_floats": int(f.value), "tiles_local": int(a.value), "tiles_total": int(b.value)}

    def set_slab_output(self, device_ptr):
        self._check(self._l.rfw_hip_set_slab_output(self._h, C.c_void_p(device_ptr)))

    def assemble_frame(self, gathered_device_ptr):
        self._check(self._l.rfw_hip_assemble_frame(self._h, C.c_void_p(gathered_device_ptr)))

    def intersect(self, origins, directions, t_min=1e-4, t_max=1e26):
        """TIntersector::intersect (crates/rfw-scene/src/intersector.rs:45-75) for a batch of rays."""
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        hits = np.empty(len(o), dtype=HIT_DTYPE)
        self._check(self._l.rfw_hip_intersect(self._h, o.ctypes.data, d.ctypes.data, t_min, t_max, len(o), hits.ctypes.data))
        return hits

    def depth_test(self, origins, directions, t_min=1e-4, t_max=1e26):
        """TIntersector::depth_test (crates/rfw-scene/src/intersector.rs:103-127): hits and BVH nodes visited per ray."""
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        hits = np.empty(len(o), dtype=HIT_DTYPE)
        depth = np.empty(len(o), dtype=np.uint32)
        self._check(self._l.rfw_hip_depth_test(self._h, o.ctypes.data, d.ctypes.data, t_min, t_max, len(o), hits.ctypes.data, depth.ctypes.data))
        return hits, depth

    def occludes_depth(self, origins, directions, t_max, t_min=1e-3):
        """occludes() plus the nodes each any-hit traversal visited (rfw_hip_debug_occludes_depth; for tools/probes)."""
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        tm = np.ascontiguousarray(t_max, dtype=np.float32)
        out = np.empty(len(o), dtype=np.uint8)
        depth = np.empty(len(o), dtype=np.uint32)
        self._check(self._l.rfw_hip_debug_occludes_depth(self._h, o.ctypes.data, d.ctypes.data, t_min, tm.ctypes.data, len(o), out.ctypes.data, depth.ctypes.data))
        return out.astype(bool), depth

    def occludes(self, origins, directions, t_max, t_min=1e-3):
        """TIntersector::occludes (crates/rfw-scene/src/intersector.rs:21-43) for a batch of rays."""
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        tm = np.ascontiguousarray(t_max, dtype=np.float32)
        out = np.empty(len(o), dtype=np.uint8)
        self._check(self._l.rfw_hip_occludes(self._h, o.ctypes.data, d.ctypes.data, t_min, tm.ctypes.data, len(o), out.ctypes.data))
        return out

    def intersect4(self, origins, directions, t_min, t_max):
        """TIntersector::intersect4 (intersector.rs:133-166): (4, 3) origins / directions, per-lane intervals -> (instance ids, prim ids, t)."""
        o = np.ascontiguousarray(np.asarray(origins, np.float32).T)   # SoA: x[4] y[4] z[4]
        d = np.ascontiguousarray(np.asarray(directions, np.float32).T)
        tmin = np.ascontiguousarray(t_min, dtype=np.float32)
        t = np.array(t_max, dtype=np.float32)
        ii, pp = np.empty(4, np.int32), np.empty(4, np.int32)
        self._check(self._l.rfw_hip_intersect4(self._h, o.ctypes.data, d.ctypes.data, tmin.ctypes.data, t.ctypes.data, ii.ctypes.data, pp.ctypes.data))
        return ii, pp, t

    def occludes4(self, origins, directions, t_min, t_max):
        o = np.ascontiguousarray(np.asarray(origins, np.float32).T)
        d = np.ascontiguousarray(np.asarray(directions, np.float32).T)
        tmin, tmax = np.ascontiguousarray(t_min, dtype=np.float32), np.ascontiguousarray(t_max, dtype=np.float32)
        out = np.empty(4, np.uint8)
        self._check(self._l.rfw_hip_occludes4(self._h, o.ctypes.data, d.ctypes.data, tmin.ctypes.data, tmax.ctypes.data, out.ctypes.data))
        return out

    def bandwidth_probe(self, nbytes=1 << 30, iterations=20):
        """Measured device copy bandwidth in GB/s (read + written bytes): the job's own HBM roofline."""
        out = C.c_double(0.0)
        self._check(self._l.rfw_hip_bandwidth_probe(self._h, nbytes, iterations, C.byref(out)))
        return float(out.value)

    def issue_probe(self, mix, trips=4000):
        """Measured vector-issue rate in G wave64 instructions / s, chip-wide, at 8 wavefronts per SIMD: mix 0 = v_fma_f32 alone, mix 1 = the
        instruction mix of the 4-wide node test (conversions, packed FMAs, min / max, compares)."""
        out = C.c_double(0.0)
        self._check(self._l.rfw_hip_issue_probe(self._h, mix, trips, C.byref(out)))
        return float(out.value)

    def lbvh_stress(self, num_boxes, iterations, seed=1):
        """(mismatches, child boxes checked) of `iterations` rebuilds of a tree over jittered boxes by the device LBVH builder."""
        err, chk = C.c_uint64(0), C.c_uint64(0)
        self._check(self._l.rfw_hip_debug_lbvh_stress(self._h, num_boxes, iterations, seed, C.byref(err), C.byref(chk)))
        return int(err.value), int(chk.value)

    def debug_read(self, what, nbytes):
        buf = np.empty(nbytes, dtype=np.uint8)
        w = C.c_uint64(0)
        self._check(self._l.rfw_hip_debug_read(self._h, what.encode(), buf.ctypes.data, nbytes, C.byref(w)))
        return buf[: int(w.value)]
