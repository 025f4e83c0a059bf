"""Incremental synchronize (gpu-rt/src/lib.rs:1345-1383): a changed mesh is rebuilt in its own region of the mega-buffers,
the other meshes are not touched; material / light edits honour the trait's `changed` bit slices and go into a new version of their
tables, so frames in flight are neither waited for nor disturbed.  Everything is checked against the oracle, which rebuilds the whole
scene every time."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = max(8, os.cpu_count() or 8)


def rays(n, seed):
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-14, 14, n), rng.uniform(0.2, 11.5, n), rng.uniform(-5.5, 5.5, n)], axis=1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # half of them aimed at the two rows of spheres (z = +-4.6, y = 0.65 / 5.6), so that edits of single spheres are seen
    k = n // 2
    tgt = np.stack([rng.uniform(-13, 13, k), rng.choice([0.65, 5.6], k), rng.choice([-4.6, 4.6], k)], axis=1).astype(np.float32)
    d[:k] = tgt - o[:k]
    d[:k] /= np.linalg.norm(d[:k], axis=1, keepdims=True)
    return o, d.astype(np.float32)


def assert_same(be, orc, o, d, view, what):
    g, r = be.intersect(o, d), orc.intersect(o, d)
    assert np.array_equal(g["inst"], r["inst"]), what
    assert np.array_equal(g["tri"], r["tri"]), what          # the boundary's triangle numbering survives the storage order
    hit = r["inst"] >= 0
    assert np.array_equal(g["t"][hit].view(np.uint32), r["t"][hit].view(np.uint32)), what
    orc.reset()
    be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), what
    assert be.scene_stats()["triangles"] == orc.stats()["n_tris"], what


def test_one_mesh_of_c4_changes_without_rebuilding_the_rest():
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 320, 180
    scene = Scene().build("atrium", 1048576, 1, 0.0, 0xC0FFEE)   # C4 with its 64 displaced icospheres as 64 meshes: 65 meshes
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    assert scene.counts()["meshes"] == 65
    be = HipBackend.init(w, h, 1.0, max_path_length=2, frames_in_flight=3)
    orc = Oracle(w, h, threads=THREADS, max_path_length=2)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o, d = rays(40000, 1)
    assert_same(be, orc, o, d, view, "full build")
    assert (be.intersect(o, d)["inst"] > 0).sum() > 2000        # rays do land on sphere instances
    nodes0 = be.scene_stats()["blas_nodes"]

    def edit(fn, what, budget_ms):
        fn()
        be.device_synchronize()
        t0 = time.perf_counter()
        scene.sync(be)                                           # set_3d_mesh of the changed meshes only + synchronize()
        host_ms = (time.perf_counter() - t0) * 1e3
        be.device_synchronize()
        total_ms = (time.perf_counter() - t0) * 1e3
        scene.mark_all_changed(); scene.sync(orc)                # the oracle rebuilds everything
        assert_same(be, orc, o, d, view, what)
        print(f"{what}: synchronize {host_ms:.2f} ms on the host, {total_ms:.2f} ms until the device is done")
        assert total_ms < budget_ms, (what, total_ms)
        return total_ms

    # 1. one sphere gets a new shape, same triangle count: rebuilt in place (the full build of this scene takes ~60 ms)
    t1 = edit(lambda: scene.replace_mesh_with_sphere(7, 6, 4242), "one mesh of 65 rebuilt in place", 4.0)
    # 2. the same again (warm: buffers and workspaces exist)
    t2 = edit(lambda: scene.replace_mesh_with_sphere(30, 29, 99), "one mesh of 65 rebuilt in place, warm", 3.0)
    # 3. a mesh that grows (5120 -> 20480 triangles) moves behind the others
    edit(lambda: scene.replace_mesh_with_sphere(12, 11, 7, quality=5), "one mesh grows and is appended", 250.0)   # (device buffers grow: allocation-bound — 25-90 ms from box to box)
    # 4. a mesh that shrinks stays where it is; a mesh is unloaded; a new mesh appears
    edit(lambda: scene.replace_mesh_with_sphere(50, 49, 3, quality=3), "one mesh shrinks in place", 8.0)
    def unload_20():
        import ctypes as C
        scene.remove_mesh(20)                                    # synchronize_system hands the unload to `be`; the oracle gets it directly
        orc._l.orc_unload_3d_meshes(orc._h, (C.c_uint32 * 1)(20), 1)
    edit(unload_20, "one mesh unloaded", 8.0)
    edit(lambda: scene.add_sphere_mesh(19, 5), "one mesh added", 250.0)
    assert be.scene_stats()["blas_nodes"] != nodes0
    be.close()


def test_material_edits_reach_later_frames_only_and_honour_changed_bits():
    """Frames in flight keep the material table they started with (a new version is written beside it); the next frames see the edit.
    More edits than table versions (4) in a row exercise the recycling of version buffers."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 160, 120
    scene = Scene().build("soup", 3000, 4, 0.0, 5)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=4)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    views = []
    for k in range(7):
        scene.set_camera([0.3 * k - 0.9, 0.4, -4.0 + 0.1 * k], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    frames = [be.host_frame() for _ in views]
    edits = [None] + [(int(rng_k.integers(0, 6)), [int(x) for x in rng_k.integers(30, 250, 3)], int(rng_k.integers(20, 250)))
                      for rng_k in [np.random.default_rng(100 + k) for k in range(1, len(views))]]
    for k, v in enumerate(views):                                # edit, synchronize, render, queue the download; never wait:
        if k:                                                    # several frames, each with ITS material table, are in flight together
            scene.recolour_material(*edits[k])
            scene.sync(be)                                       # set_materials(changed = one bit) + synchronize
        be.render(v)
        be.download_frame(frames[k], accumulator=True)
    # the oracle replays the same edits on its own copy of the scene
    scene_o = Scene().build("soup", 3000, 4, 0.0, 5)
    scene_o.set_aspect(w / h)
    want = []
    for k, v in enumerate(views):
        if k:
            scene_o.recolour_material(*edits[k])
        scene_o.mark_all_changed(); scene_o.sync(orc)
        orc.reset(); orc.render(v)
        want.append(orc.accumulator().copy())
    be.wait_downloads()
    for k in range(len(views)):
        assert np.array_equal(frames[k].view(np.uint32), want[k].view(np.uint32)), k
    assert not np.array_equal(want[0], want[-1])
    be.close()


def test_material_edit_before_every_render_with_two_frame_slots():
    """with two frame slots and an edit before EVERY render a slot's older frame may still be executing when the upload of
    version v + 4 recycles the buffer it reads while the slot's LATEST frame already reads a newer version.  The recycle check must look at
    the oldest frame of a slot that may still run, not at the latest.  A larger image makes the frames long against the host's edit loop."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 640, 360
    scene = Scene().build("soup", 20000, 4, 0.0, 6)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=2)
    orc = Oracle(w, h, threads=THREADS, max_path_length=3)
    scene.sync(be)
    n_frames = 14
    views = []
    for k in range(n_frames):
        scene.set_camera([0.2 * k - 1.2, 0.4, -4.0 + 0.05 * k], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    frames = [be.host_frame() for _ in views]
    edits = [None] + [(int(r.integers(0, 6)), [int(x) for x in r.integers(30, 250, 3)], int(r.integers(20, 250)))
                      for r in [np.random.default_rng(700 + k) for k in range(1, n_frames)]]
    for k, v in enumerate(views):                                # never wait: edits race ahead of the frames
        if k:
            scene.recolour_material(*edits[k])
            scene.sync(be)
        be.render(v)
        be.download_frame(frames[k], accumulator=True)
    scene_o = Scene().build("soup", 20000, 4, 0.0, 6)
    scene_o.set_aspect(w / h)
    be.wait_downloads()
    for k, v in enumerate(views):
        if k:
            scene_o.recolour_material(*edits[k])
        scene_o.mark_all_changed(); scene_o.sync(orc)
        orc.reset(); orc.render(v)
        assert np.array_equal(frames[k].view(np.uint32), orc.accumulator().view(np.uint32)), k
    be.close()


def test_texture_edits_honour_changed_bits():
    """VERDICT r01 #13, textures: set_textures with the trait's `changed` bit slice copies and resamples (gpu-rt's 1024 x 1024 x 5 array)
    only the textures whose bit is set and synchronize() writes only those over the old texels; images before and after every edit equal
    the oracle's, which takes everything again each time.  A change of the count, or of the skybox, lays the whole array out again."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 160, 104
    scene = Scene().build("gallery", 0, 0, 0.0, 5)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=3)
    orc = Oracle(w, h, threads=THREADS, max_path_length=3)

    def both(what):
        scene.sync(be)                       # carries the bit slice of what was edited
        scene.mark_all_changed(); scene.sync(orc)
        be.reset_accumulation(); orc.reset()
        for _ in range(2):
            be.render(view); orc.render(view)
        ga, ra = be.accumulator(), orc.accumulator()
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), what
        return ga.copy()

    scene.mark_all_changed()
    first = both("first upload")
    n_tex = 0
    while True:
        try:
            scene.texture(n_tex); n_tex += 1
        except KeyError:
            break
    assert n_tex >= 2
    t_full = time.perf_counter(); scene.mark_all_changed(); scene.sync(be); t_full = time.perf_counter() - t_full
    seen = [first]
    for k in range(n_tex):                   # every texture once, one at a time
        scene.repaint_texture(k, 100 + k)
        seen.append(both(f"texture {k} repainted"))
    assert any(not np.array_equal(seen[0], s) for s in seen[1:])
    scene.repaint_texture(0, 7); scene.repaint_texture(n_tex - 1, 8)     # two at once
    t_part = time.perf_counter(); scene.sync(be); t_part = time.perf_counter() - t_part
    scene.mark_all_changed(); scene.sync(orc)
    be.reset_accumulation(); orc.reset()
    be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    if n_tex >= 4:
        assert t_part < t_full, (t_part, t_full)                          # two of n textures resampled and uploaded, not all
    be.close()


def test_edits_and_skinned_refits_without_the_packet_form_of_the_node_copies():
    """with option packet_trace = 0 chosen before the build the packet form of the per-octant node copies is never
    allocated; an incremental edit of a mesh whose nodes do not start at 0, and the per-frame refit of a skinned copy (which lies behind the
    static meshes), must then leave that absent buffer alone — both used to hand the expand kernel `nullptr + node_base`."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 160, 96
    scene = Scene().build("atrium", 620000, 1, 0.0, 0xBEEF)     # (spheres exist above 600 000 triangles) the spheres as meshes of their own: every mesh but the first has node_base > 0
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    assert scene.counts()["meshes"] > 8
    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    be.set_option("packet_trace", 0)                            # before the first synchronize: no packet copies
    orc = Oracle(w, h, threads=THREADS, max_path_length=2)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    assert be.scene_stats()["packet_copies"] == 0
    o, d = rays(20000, 3)
    assert_same(be, orc, o, d, view, "full build without packet copies")
    for k, (mesh, seed) in enumerate(((7, 11), (3, 12), (7, 13))):
        scene.replace_mesh_with_sphere(mesh, mesh - 1, seed)    # incremental: rebuilt in its own region (node_base > 0)
        scene.sync(be)
        scene.mark_all_changed(); scene.sync(orc)
        assert be.scene_stats()["packet_copies"] == 0
        assert_same(be, orc, o, d, view, f"edit {k} without packet copies")
    be.close()

    # a skinned copy is refitted every synchronize, behind the static meshes
    scene = Scene().build("cornell")
    scene.build("skinned", 0, 0, 0.0, 3)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    be.set_option("packet_trace", 0)
    orc = Oracle(w, h, threads=THREADS, max_path_length=2)
    for t in (0.3, 1.1, 0.0, 1.7):
        scene.pose(t)
        scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
        assert be.scene_stats()["packet_copies"] == 0
        orc.reset()
        be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), t
    be.close()
