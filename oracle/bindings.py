"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from rfw_rs_amd import pod  # noqa: E402  (POD struct mirrors only: shared boundary definitions)
from rfw_rs_amd.scene import BackendTable  # noqa: E402

ORACLE_LIB = os.path.join(_HERE, "liboracle.so")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary", "extension", "shadow", "top_nodes", "mesh_nodes", "tris", "instances",
                                          "n_tris", "n_instances", "n_mesh_mbvh_nodes", "n_top_mbvh_nodes")] + [("sample_count", C.c_uint32), ("pad", C.c_uint32)]


_libs = {}


def build_timing_library():
    """The oracle's source compiled again for TIMING on the machine that runs the bench: -O3 -march=native (code for this CPU does not
    travel, so it is built where it runs, into a temporary directory), still -ffp-contract=off, so its images are the checker's images.
    Returns (path, flags) or (None, reason)."""
    import hashlib
    import subprocess
    import tempfile
    src = os.path.join(_HERE, "oracle.cpp")
    flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-pthread", "-shared"]
    tag = hashlib.sha1(open(src, "rb").read() + " ".join(flags).encode()).hexdigest()[:12]
    out_dir = os.path.join(tempfile.gettempdir(), f"rfw_oracle_timing_{os.getuid()}")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"liboracle_timing_{tag}.so")
    if not os.path.exists(out):
        try:
            r = subprocess.run([os.environ.get("CXX", "g++")] + flags + ["-o", out + ".tmp", src], capture_output=True, text=True, timeout=300)
        except Exception as e:  # no compiler on this machine
            return None, f"timing build failed ({e})"
        if r.returncode != 0:
            return None, "timing build failed: " + r.stderr[-200:]
        os.replace(out + ".tmp", out)
    return out, " ".join(flags[:2] + flags[4:5])


def lib(path=None):
    path = path or ORACLE_LIB
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        l = C.CDLL(path)
        l.orc_create.restype = C.c_void_p
        l.orc_create.argtypes = [C.c_uint32, C.c_uint32]
        l.orc_destroy.argtypes = [C.c_void_p]
        l.orc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        l.orc_synchronize.argtypes = [C.c_void_p]
        l.orc_render.argtypes = [C.c_void_p, C.POINTER(pod.CameraView3D)]
        l.orc_reset.argtypes = [C.c_void_p]
        l.orc_read_accumulator.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        l.orc_read_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        l.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_int]
        l.orc_occludes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        l.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(OrcStats)]
        l.orc_generate_primary_rays.argtypes = [C.c_void_p, C.POINTER(pod.CameraView3D), C.c_uint32, C.c_void_p, C.c_void_p]
        l.orc_validate_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        l.orc_detmath_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        l.orc_sample_texture.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]
        l.orc_set_blue_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        l.orc_blue_noise_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
        l.orc_blue_noise_sample.restype = C.c_float
        l.orc_eval_shading.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        l.orc_wang_hash.argtypes = [C.c_uint32]
        l.orc_wang_hash.restype = C.c_uint32
        l.orc_randi.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_randi.restype = C.c_uint32
        l.orc_randf.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_randf.restype = C.c_float
        l.orc_pack_normal.argtypes = [C.c_float, C.c_float, C.c_float]
        l.orc_pack_normal.restype = C.c_uint32
        l.orc_unpack_normal.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        l.orc_safe_origin.argtypes = [C.POINTER(C.c_float)] * 4
        l.orc_intersect_triangle.argtypes = [C.POINTER(pod.RTTriangle), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]
        l.orc_mat4_inverse.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _libs[path] = l
    return _libs[path]


HIT_DTYPE = np.dtype([("inst", "<i4"), ("tri", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


class Oracle:
    def __init__(self, width, height, library=None, **options):
        self._l = lib(library)
        self.width, self.height = width, height
        self._h = C.c_void_p(self._l.orc_create(width, height))
        for k, v in options.items():
            self.set_option(k, v)

    def __del__(self):
        try:
            self._l.orc_destroy(self._h)
        except Exception:
            pass

    def last_error(self):
        return "oracle error"

    def table(self):
        t = BackendTable()
        t.instance = self._h
        for name in ("set_3d_mesh", "unload_3d_meshes", "set_3d_instances", "set_materials", "synchronize",
                     "set_point_lights", "set_spot_lights", "set_area_lights", "set_directional_lights", "set_textures", "set_skybox", "set_skins"):
            setattr(t, name, C.cast(getattr(self._l, "orc_" + name), C.c_void_p))
        return t

    def set_option(self, key, value):
        if self._l.orc_set_option(self._h, key.encode(), float(value)) != 0:
            raise KeyError(key)

    def set_blue_noise(self, table):
        if table is None:
            assert self._l.orc_set_blue_noise(self._h, None, 0) == 0
        else:
            t = np.ascontiguousarray(table, dtype=np.uint32)
            assert self._l.orc_set_blue_noise(self._h, t.ctypes.data, t.size) == 0

    def blue_noise_sample(self, sample_count, x, y, dim):
        return float(self._l.orc_blue_noise_sample(self._h, sample_count, x, y, dim))

    def eval_shading(self, op, inputs):
        a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 48)
        out = np.zeros((len(a), 12), dtype=np.float32)
        assert self._l.orc_eval_shading(self._h, op, len(a), a.ctypes.data, out.ctypes.data) == 0
        return out

    def render(self, view):
        self._l.orc_render(self._h, C.byref(view))

    def reset(self):
        self._l.orc_reset(self._h)

    def accumulator(self):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        assert self._l.orc_read_accumulator(self._h, a.ctypes.data, a.size) == 0
        return a

    def framebuffer(self):
        a = np.empty((self.height, self.width, 4), dtype=np.float32)
        assert self._l.orc_read_framebuffer(self._h, a.ctypes.data, a.size) == 0
        return a

    def intersect(self, origins, directions, t_min=1e-4, t_max=1e26, brute=False):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        hits = np.empty(len(o), dtype=HIT_DTYPE)
        self._l.orc_intersect(self._h, o.ctypes.data, d.ctypes.data, t_min, t_max, len(o), hits.ctypes.data, 1 if brute else 0)
        return hits

    def occludes(self, origins, directions, t_max, t_min=1e-3, brute=False):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        tm = np.ascontiguousarray(t_max, dtype=np.float32)
        out = np.empty(len(o), dtype=np.uint8)
        self._l.orc_occludes(self._h, o.ctypes.data, d.ctypes.data, t_min, tm.ctypes.data, len(o), out.ctypes.data, 1 if brute else 0)
        return out

    def primary_rays(self, view, sample=0):
        n = self.width * self.height
        o = np.empty((n, 3), dtype=np.float32)
        d = np.empty((n, 3), dtype=np.float32)
        self._l.orc_generate_primary_rays(self._h, C.byref(view), sample, o.ctypes.data, d.ctypes.data)
        return o, d

    def sample_texture(self, tex, u, v, lod, trilinear=False):
        out = (C.c_float * 4)()
        self._l.orc_sample_texture(self._h, tex, u, v, lod, 1 if trilinear else 0, out)
        return np.array(out[:], dtype=np.float32)

    def triangles(self):
        """All triangles after synchronize as raw 176-byte records (n, 44) float32: static meshes, then skinned copies."""
        self._l.orc_read_triangles.restype = C.c_uint64
        self._l.orc_read_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        n = int(self._l.orc_read_triangles(self._h, None, 0))
        out = np.zeros((n, 44), dtype=np.float32)
        self._l.orc_read_triangles(self._h, out.ctypes.data, n)
        return out

    def stats(self):
        s = OrcStats()
        self._l.orc_get_stats(self._h, C.byref(s))
        d = {n: getattr(s, n) for n, _ in OrcStats._fields_ if n != "pad"}
        d["busy_threads"] = int(s.pad)  # threads that rendered at least one tile of the last frame
        return d

    def validate_bvh(self):
        e = C.c_uint64(0)
        self._l.orc_validate_bvh(self._h, C.byref(e))
        return int(e.value)


def detmath(fn, x, y=None):
    names = {"sin": 0, "cos": 1, "log": 2, "exp": 3, "acos": 4, "atan2": 5, "log2": 6, "asin": 7}
    x = np.ascontiguousarray(x, dtype=np.float32)
    y2 = np.ascontiguousarray(y if y is not None else x, dtype=np.float32)
    out = np.empty_like(x)
    assert lib().orc_detmath_eval(names[fn], x.ctypes.data, y2.ctypes.data, out.ctypes.data, x.size) == 0
    return out
