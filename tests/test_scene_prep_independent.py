"""Scene preparation held against an INDEPENDENT restatement.

Every GPU-vs-oracle parity test feeds both sides from the same C++ host library (rfw-rs_amd/host/librfw_host.so), so a slip in its
`Mesh3D::from`, `into_device_material` or `update_lights` would be invisible to all of them.  Here a small glTF document (16 triangles in
three meshes, four instances, one emissive material) goes through the host library into a backend that only RECORDS what the trait calls
hand it, and the recorded `RTTriangle` / `DeviceMaterial` / `AreaLight` bytes are compared with arrays built in numpy straight from the
reference's formulas:
    crates/rfw-scene/src/objects_3d/mod.rs:673-850   Mesh3D::from (normals when the document has none, lod, area, ids)
    crates/rfw-backend/src/structs.rs:970-984         RTTriangle::normal, RTTriangle::area (Heron)
    crates/rfw-scene/src/material/list.rs:755-814     into_device_material (16 parameters as bytes, low byte first; flags; map ids)
    crates/rfw-scene/src/lib.rs:575-648               update_lights (an AreaLight per emissive triangle and instance, light ids written back)
    crates/rfw-backend/src/lights.rs:70-97            AreaLight::new
Integers, bytes and ids must agree exactly; floats to 2 ulp (the two sides are free to order their additions differently)."""
import ctypes as C

import numpy as np

from rfw_rs_amd import Scene, pod
from rfw_rs_amd.scene import BackendTable

import gltf_util

F = np.float32


class Recorder:
    """A `Backend` that keeps copies of what the trait calls borrow (crates/rfw-backend/src/lib.rs:36-81)."""

    def __init__(self):
        self.meshes, self.instances, self.materials, self.area_lights = {}, {}, None, None
        vp, u32 = C.c_void_p, C.c_uint32
        self._cbs = []

        def cb(restype, *argtypes):
            def deco(fn):
                f = C.CFUNCTYPE(restype, *argtypes)(fn)
                self._cbs.append(f)
                return f
            return deco

        @cb(C.c_int, vp, u32, C.POINTER(pod.MeshData3D))
        def set_3d_mesh(_, mesh_id, d):
            d = d.contents
            tris = np.frombuffer(C.string_at(d.triangles, d.num_triangles * 176), dtype=np.uint8).reshape(d.num_triangles, 176).copy()
            verts = np.frombuffer(C.string_at(d.vertices, d.num_vertices * 64), dtype=np.uint8).reshape(d.num_vertices, 64).copy()
            ranges = [(r.first, r.last, r.mat_id) for r in (d.ranges[k] for k in range(d.num_ranges))]
            self.meshes[mesh_id] = {"triangles": tris, "vertices": verts, "ranges": ranges}
            return 0

        @cb(C.c_int, vp, u32, C.POINTER(pod.InstancesData3D))
        def set_3d_instances(_, mesh_id, d):
            d = d.contents
            self.instances[mesh_id] = np.frombuffer(C.string_at(d.matrices, d.num_matrices * 64), dtype=np.float32).reshape(d.num_matrices, 16).copy()
            return 0

        @cb(C.c_int, vp, vp, u32, vp)
        def set_materials(_, ptr, n, changed):
            self.materials = np.frombuffer(C.string_at(ptr, n * 96), dtype=np.uint8).reshape(n, 96).copy()
            return 0

        @cb(C.c_int, vp, vp, u32, vp)
        def set_area_lights(_, ptr, n, changed):
            self.area_lights = np.frombuffer(C.string_at(ptr, n * 96), dtype=np.uint8).reshape(n, 96).copy() if n else np.zeros((0, 96), np.uint8)
            return 0

        @cb(C.c_int, vp, vp, u32, vp)
        def ignore4(_, a, n, c):
            return 0

        @cb(C.c_int, vp, vp, u32)
        def ignore3(_, a, n):
            return 0

        @cb(C.c_int, vp, vp)
        def ignore2(_, a):
            return 0

        @cb(C.c_int, vp)
        def ignore1(_):
            return 0

        t = BackendTable()
        t.instance = None
        cast = lambda f: C.cast(f, C.c_void_p)
        t.set_3d_mesh, t.set_3d_instances, t.set_materials, t.set_area_lights = cast(set_3d_mesh), cast(set_3d_instances), cast(set_materials), cast(set_area_lights)
        t.unload_3d_meshes = cast(ignore3)
        t.synchronize = cast(ignore1)
        t.set_point_lights = t.set_spot_lights = t.set_directional_lights = t.set_textures = t.set_skins = cast(ignore4)
        t.set_skybox = cast(ignore2)
        self._table = t

    def table(self):
        return self._table

    def last_error(self):
        return "recorder"


# ---- the reference's formulas in float32, one operation at a time
def sub(a, b): return (a - b).astype(F)
def cross(a, b): return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)
def length(a): return F(np.sqrt(F(F(F(a[0] * a[0]) + F(a[1] * a[1])) + F(a[2] * a[2]))))
def normalize(a): return (a * F(F(1.0) / length(a))).astype(F)                      # glam: v * length_recip()


def tri_normal(v0, v1, v2):                                                         # structs.rs:970-975
    return normalize(cross(sub(v1, v0), sub(v2, v0)))


def tri_area(v0, v1, v2):                                                           # structs.rs:977-984 (Heron)
    a, b, c = length(sub(v1, v0)), length(sub(v2, v1)), length(sub(v0, v2))
    s = F(F(F(a + b) + c) * F(0.5))
    return F(np.sqrt(F(F(F(s * F(s - a)) * F(s - b)) * F(s - c))))


def lod_of(v0, v1, v2, uv0, uv1, uv2):                                              # objects_3d/mod.rs:820-823
    ta = F(F(1024 * 1024) * abs(F(F(F(uv1[0] - uv0[0]) * F(uv2[1] - uv0[1])) - F(F(uv2[0] - uv0[0]) * F(uv1[1] - uv0[1])))))
    pa = length(cross(sub(v1, v0), sub(v2, v0)))
    with np.errstate(divide="ignore", invalid="ignore"):
        x = F(np.sqrt(F(F(0.5) * F(np.log2(F(ta / pa))))))
    return F(0.0) if np.isnan(x) else max(F(0.0), x)                                # 0.0_f32.max(NaN) = 0.0


def ulp_close(a, b, ulps=2):
    a, b = np.asarray(a, F), np.asarray(b, F)
    return np.all(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)) <= ulps) or np.allclose(a, b, rtol=0, atol=1e-7)


def document_triangles(doc, blob, mesh_index):
    """De-indexed positions / normals / uvs of one glTF mesh (what l3d hands Mesh3D::from as a MeshDescriptor), in document order."""
    def acc(i):
        a = doc["accessors"][i]
        v = doc["bufferViews"][a["bufferView"]]
        dt = {5126: np.float32, 5123: np.uint16, 5125: np.uint32}[a["componentType"]]
        w = {"SCALAR": 1, "VEC2": 2, "VEC3": 3}[a["type"]]
        return np.frombuffer(blob, dtype=dt, count=a["count"] * w, offset=v["byteOffset"]).reshape(a["count"], w)
    pos, nor, uv, mats = [], [], [], []
    for prim in doc["meshes"][mesh_index]["primitives"]:
        p = acc(prim["attributes"]["POSITION"])
        idx = acc(prim["indices"]).ravel() if "indices" in prim else np.arange(len(p))
        pos.append(p[idx])
        nor.append(acc(prim["attributes"]["NORMAL"])[idx] if "NORMAL" in prim["attributes"] else np.zeros((len(idx), 3), F))
        uv.append(acc(prim["attributes"]["TEXCOORD_0"])[idx] if "TEXCOORD_0" in prim["attributes"] else np.zeros((len(idx), 2), F))
        mats += [prim["material"]] * (len(idx) // 3)
    return np.concatenate(pos).astype(F), np.concatenate(nor).astype(F), np.concatenate(uv).astype(F), mats


def test_triangles_materials_and_area_lights_match_an_independent_restatement(tmp_path):
    path = gltf_util.write_gltf(tmp_path)
    doc, blob = gltf_util.build_document()
    scene = Scene().load_gltf(str(path))
    rec = Recorder()
    scene.sync(rec)
    assert sorted(rec.meshes) == [0, 1, 2] and sum(len(m["triangles"]) for m in rec.meshes.values()) == 16

    # ---- RTTriangle records (Mesh3D::from)
    tri_dt = np.dtype([("v0", F, 3), ("u0", F), ("v1", F, 3), ("u1", F), ("v2", F, 3), ("u2", F), ("normal", F, 3), ("vv0", F), ("n0", F, 3), ("vv1", F),
                       ("n1", F, 3), ("vv2", F), ("n2", F, 3), ("id", np.int32), ("t0", F, 4), ("t1", F, 4), ("t2", F, 4), ("light_id", np.int32),
                       ("mat_id", np.int32), ("lod", F), ("area", F)])
    assert tri_dt.itemsize == 176
    scene_material_of = {}                                           # document material -> scene material id, learnt from the records themselves
    for mesh_id, mesh in rec.meshes.items():
        pos, nor, uv, mats = document_triangles(doc, blob, mesh_id)
        t = mesh["triangles"].view(tri_dt).ravel()
        assert len(t) == len(pos) // 3
        has_normals = bool(np.any(nor[0] != 0))                     # objects_3d/mod.rs:680: the FIRST normal decides
        for i in range(len(t)):
            v0, v1, v2 = pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]
            assert np.array_equal(t["v0"][i], v0) and np.array_equal(t["v1"][i], v1) and np.array_equal(t["v2"][i], v2)
            assert (t["u0"][i], t["u1"][i], t["u2"][i]) == (uv[3 * i][0], uv[3 * i + 1][0], uv[3 * i + 2][0])
            assert (t["vv0"][i], t["vv1"][i], t["vv2"][i]) == (uv[3 * i][1], uv[3 * i + 1][1], uv[3 * i + 2][1])
            assert ulp_close(t["normal"][i], tri_normal(v0, v1, v2))
            if has_normals:
                want_n = nor[3 * i:3 * i + 3]
            else:                                                    # every de-indexed vertex belongs to this triangle only: its normal is the
                want_n = np.tile(normalize((tri_normal(v0, v1, v2) * tri_area(v0, v1, v2)).astype(F)), (3, 1))   # area-weighted face normal, normalised
            assert ulp_close(np.stack([t["n0"][i], t["n1"][i], t["n2"][i]]), want_n, ulps=4)
            assert t["id"][i] == i
            assert ulp_close(t["area"][i], tri_area(v0, v1, v2), ulps=4)
            assert ulp_close(t["lod"][i], lod_of(v0, v1, v2, uv[3 * i], uv[3 * i + 1], uv[3 * i + 2]), ulps=4)
            scene_material_of.setdefault(mats[i], int(t["mat_id"][i]))
            assert int(t["mat_id"][i]) == scene_material_of[mats[i]]
        # ranges: one VertexMesh per run of equal material ids; first / last are VERTEX indices (objects_3d/mod.rs:751-757)
        assert mesh["ranges"] == [(0, len(pos), scene_material_of[mats[0]])]

    # ---- DeviceMaterial records (into_device_material) from the Material the host derived
    mat_dt = np.dtype([("color", F, 4), ("absorption", F, 4), ("specular", F, 4), ("parameters", np.uint8, 16), ("flags", np.uint32), ("maps", np.int32, 5), ("pad", np.int32, 2)])
    assert mat_dt.itemsize == 96
    dm = rec.materials.view(mat_dt).ravel()
    names = ["metallic", "subsurface", "specular_f", "roughness", "specular_tint", "anisotropic", "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss", "transmission", "eta"]
    for k in range(len(dm)):
        m = scene.material(k)
        to_char = lambda f: int(min(F(F(f) * F(255.0)), F(255.0)))     # `(f * 255.0).min(255.0) as u8`: truncation, saturating
        want = [to_char(m[n]) for n in names]
        assert list(dm["parameters"][k][:12]) == want, (k, list(dm["parameters"][k]), want)
        assert np.array_equal(dm["color"][k], np.array(m["color"], F)) and np.array_equal(dm["specular"][k], np.array(m["specular"], F))
        maps = [m["diffuse_tex"], m["normal_tex"], m["metallic_roughness_tex"], m["emissive_tex"], m["sheen_tex"]]
        assert list(dm["maps"][k]) == maps
        flags = (1 if maps[0] >= 0 else 0) | (2 if maps[1] >= 0 else 0) | (12 if maps[2] >= 0 else 0) | (16 if maps[3] >= 0 else 0) | (32 if maps[4] >= 0 else 0)
        assert int(dm["flags"][k]) == flags
    # the document's materials arrived as the document says (pbrMetallicRoughness; emissive factor x strength as a colour above 1)
    red, grey, emitter = (scene.material(scene_material_of[j]) for j in range(3))
    assert np.allclose(red["color"][:3], [0.8, 0.1, 0.1]) and abs(red["roughness"] - 0.6) < 1e-6 and red["metallic"] == 0.0
    assert np.allclose(grey["color"][:3], [0.6, 0.6, 0.6]) and abs(grey["roughness"] - 0.9) < 1e-6
    assert np.allclose(emitter["color"][:3], [12.0, 10.8, 9.6], rtol=1e-6)

    # ---- AreaLight records (update_lights): one per triangle of an emissive range and instance, in mesh / instance / triangle order
    al_dt = np.dtype([("position", F, 3), ("energy", F), ("normal", F, 3), ("area", F), ("vertex0", F, 3), ("inst_idx", np.int32), ("vertex1", F, 3),
                      ("mesh_id", np.int32), ("radiance", F, 3), ("d1", np.int32), ("vertex2", F, 3), ("d2", np.int32)])
    assert al_dt.itemsize == 96
    lights = rec.area_lights.view(al_dt).ravel()
    want, base = [], 0
    for mesh_id in sorted(rec.meshes):
        pos, _, _, mats = document_triangles(doc, blob, mesh_id)
        for slot, mtx in enumerate(rec.instances[mesh_id]):
            M = mtx.reshape(4, 4).T.astype(F)                      # column-major
            for i, mat in enumerate(mats):
                if max(scene.material(scene_material_of[mat])["color"][:3]) <= 1.0:
                    continue
                w = []
                for v in pos[3 * i:3 * i + 3]:                      # transform * vertex, column sum ((c0 x + c1 y) + c2 z) + c3
                    w.append((((M[:3, 0] * v[0]).astype(F) + (M[:3, 1] * v[1]).astype(F)).astype(F) + (M[:3, 2] * v[2]).astype(F)).astype(F) + M[:3, 3])
                w = [x.astype(F) for x in w]
                col = np.abs(np.array(scene.material(scene_material_of[mat])["color"][:3], F))
                want.append({"position": ((w[0] + w[1]).astype(F) + w[2]).astype(F) * F(F(1.0) / F(3.0)), "energy": length(col), "normal": tri_normal(*w), "area": tri_area(*w),
                             "v": w, "inst": base + slot, "mesh": mesh_id, "radiance": col, "tri": (mesh_id, i)})
        base += len(rec.instances[mesh_id])
    assert len(lights) == len(want) == 2                             # the lamp's two triangles, one instance
    for k, wl in enumerate(want):
        L = lights[k]
        assert ulp_close(L["position"], wl["position"], 4) and ulp_close(L["energy"], wl["energy"], 4) and ulp_close(L["normal"], wl["normal"], 4)
        assert ulp_close(L["area"], wl["area"], 4) and np.array_equal(L["radiance"], wl["radiance"])
        assert ulp_close(L["vertex0"], wl["v"][0]) and ulp_close(L["vertex1"], wl["v"][1]) and ulp_close(L["vertex2"], wl["v"][2])
        assert (int(L["inst_idx"]), int(L["mesh_id"]), int(L["d1"]), int(L["d2"])) == (wl["inst"], wl["mesh"], 1, 2)
        mesh_id, i = wl["tri"]
        assert int(rec.meshes[mesh_id]["triangles"].view(tri_dt).ravel()["light_id"][i]) == k       # lib.rs:640-646: ids written back
    for mesh_id, mesh in rec.meshes.items():                         # every other triangle: no light
        ids = mesh["triangles"].view(tri_dt).ravel()["light_id"]
        assert sorted(int(x) for x in ids if x >= 0) == ([0, 1] if mesh_id == want[0]["mesh"] else [])
    assert np.allclose(lights["normal"], [[0, -1, 0]] * 2, atol=1e-6)   # the lamp faces down


def test_into_device_material_bytes_on_awkward_values():
    """to_char = `(f * 255.0).min(255.0) as u8`: truncation (not rounding), saturation above 1, and `as u8` of a negative or NaN float is 0."""
    from rfw_rs_amd.scene import host_lib
    l = host_lib()
    l.rfwhost_into_device_material.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(pod.DeviceMaterial)]
    vals = [0.0, 1.0, 0.5, 0.999, 1.0 / 255.0, 0.00392, 2.5, -0.25, 0.3333, 0.6, 0.9, float("nan"), 0.25, 0.75, 0.125, 0.0625]
    color = (C.c_float * 4)(0.25, 0.5, 0.75, 1.0)
    params = (C.c_float * 16)(*vals)
    out = pod.DeviceMaterial()
    assert l.rfwhost_into_device_material(color, params, C.byref(out)) == 0
    got = np.frombuffer(bytes(out.parameters), dtype=np.uint8)

    def to_char(f):
        x = F(F(f) * F(255.0))
        x = F(255.0) if np.isnan(x) else min(x, F(255.0))          # f32::min returns the other operand for a NaN
        return 0 if x <= 0 else int(x)                              # `as u8` saturates
    assert list(got) == [to_char(v) for v in vals]
