"""The C++ host mirror of the rfw-scene pieces that define the backend's inputs."""
import numpy as np
import pytest

from rfw_rs_amd import Scene


def tri_array(md):
    n = md.num_triangles
    buf = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_float * (44 * n)).from_address(np.ctypeslib.ctypes.addressof(md.triangles.contents)))
    return buf.reshape(n, 44).copy()


def test_cornell_has_36_triangles_and_two_area_lights():
    s = Scene().build("cornell")
    c = s.counts()
    assert s.triangle_count == 36 and c["materials"] == 4 and c["area_lights"] == 2 and c["instances"] == 1
    t = tri_array(s.mesh_data(0))
    v0, v1, v2, gn = t[:, 0:3], t[:, 4:7], t[:, 8:11], t[:, 12:15]
    n = np.cross(v1 - v0, v2 - v0)
    area = 0.5 * np.linalg.norm(n, axis=1)
    assert np.allclose(gn, n / np.linalg.norm(n, axis=1, keepdims=True), atol=1e-5)     # RTTriangle::normal (structs.rs:970-975)
    assert np.allclose(t[:, 43], area, rtol=1e-4)                                        # RTTriangle::area (Heron, structs.rs:977-984)
    light_id = t[:, 40].view(np.int32)
    assert sorted(light_id[light_id >= 0]) == [0, 1]                                     # update_lights wrote ids back (lib.rs:640-646)
    assert np.allclose(gn[light_id >= 0], [0, -1, 0], atol=1e-6)                          # the lamp faces down into the box
    assert (t[:, 27].view(np.int32) == np.arange(36)).all()                              # id = triangle index


def test_atrium_hits_target_triangle_count():
    for target in (20000, 262267):
        s = Scene().build("atrium", target, 0, 0.0, 0xC0FFEE)
        assert abs(s.triangle_count - target) / target < (0.05 if target < 100000 else 0.01)
        c = s.counts()
        assert c["materials"] == 25 and c["area_lights"] == 4 and c["directional_lights"] == 1


def test_sphere_grid_animation_moves_instances():
    s = Scene().build("cornell").build("spheres", 4, 4, 1.0)
    assert s.counts()["instances"] == 17 and s.triangle_count == 36 + 16 * 320
    s.animate(0.5)


def test_skinned_scene_poses_change_only_the_skinned_copies():
    """build_skinned: a floor, a bind-pose tube and two instances of the same tube wearing different skins."""
    import numpy as np
    from oracle.bindings import Oracle
    from rfw_rs_amd import Scene
    scene = Scene().build("skinned", 0, 0, 0.0, 3)
    orc = Oracle(32, 24, threads=2)
    scene.sync(orc)
    assert orc.stats()["n_tris"] == 6 + 448 + 2 * 448 and orc.validate_bvh() == 0
    a = orc.triangles().copy()
    scene.pose(1.3)
    scene.mark_all_changed()
    scene.sync(orc)
    b = orc.triangles()
    assert orc.validate_bvh() == 0
    assert np.array_equal(a[:454].view(np.uint32), b[:454].view(np.uint32))      # static meshes: unchanged
    assert not np.array_equal(a[454:], b[454:])                                  # skinned copies: moved
    assert not np.array_equal(b[454:902], b[902:])                               # two skins, two shapes


def test_quad3d_matches_the_reference_construction():
    """objects_3d/quad.rs:52-75: tangent = 0.5 w normalize(n x helper), bitangent = 0.5 h normalize(tangent) x n; two triangles, the given
    normal at every vertex; an emissive material makes it two area lights."""
    from oracle.bindings import Oracle
    scene = Scene().build("cornell")
    first = scene.counts()
    light = next(i for i in range(first["materials"]) if max(scene.material(i)["color"][:3]) > 1.0)    # the Cornell box's emitter
    mesh = scene.add_quad((0.0, 2.0, 0.0), (0.5, 1.0, -0.25), 3.0, 2.0, light)
    assert scene.counts()["meshes"] == first["meshes"] + 1 and scene.counts()["area_lights"] == first["area_lights"] + 2
    orc = Oracle(16, 16, threads=1)
    scene.sync(orc)
    tris = orc.triangles()[-2:]
    n = np.array([0.0, 1.0, 0.0]); pos = np.array([0.5, 1.0, -0.25])
    t = np.cross(n, [1.0, 0.0, 0.0]); t = 1.5 * t / np.linalg.norm(t)
    b = 1.0 * np.cross(t / np.linalg.norm(t), n)
    want = [pos - b - t, pos + b - t, pos - b + t, pos + b - t, pos + b + t, pos - b + t]
    got = np.concatenate([tris[:, [0, 1, 2]], tris[:, [4, 5, 6]], tris[:, [8, 9, 10]]], axis=1).reshape(2, 3, 3).reshape(6, 3)
    assert np.allclose(got, np.array(want, np.float32), atol=1e-6)
    assert np.allclose(tris[:, 16:19], n)                                       # the vertex normals are the given normal, normalised
    with pytest.raises(ValueError):
        scene.add_quad((0, 1, 0), (0, 0, 0), 1.0, 1.0, 999)


def test_camera_moves_follow_the_reference():
    """camera/mod.rs:164-186 in the frame of calculate_matrix (:246-252): x = normalize(z x Y), y = normalize(x x z)."""
    scene = Scene().build("cornell")
    pos, d = scene.look_at((1.0, 2.0, 3.0), (1.0, 2.0, 7.0))
    assert pos == [1.0, 2.0, 3.0] and d == [0.0, 0.0, 1.0]
    pos, d = scene.translate_relative((0.5, 0.25, 2.0))             # x = z x Y = (-1, 0, 0) for z = +Z
    assert np.allclose(pos, [0.5, 2.25, 5.0]) and d == [0.0, 0.0, 1.0]
    pos, d = scene.translate_target((1.0, 0.0, 0.0))                # towards the camera's right
    assert np.allclose(d, np.array([-1.0, 0.0, 1.0]) / np.sqrt(2.0), atol=1e-7) and np.allclose(pos, [0.5, 2.25, 5.0])
    v = scene.view(64, 32)
    assert np.allclose([v.direction.x, v.direction.y, v.direction.z], d, atol=1e-7)
