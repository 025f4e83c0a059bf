"""The C++ host mirror of the rfw-scene pieces that define the backend's inputs."""
import numpy as np

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
