"""Behaviour of the drop-in boundary on a real device: the call sequences the reference's host makes
(rfw/src/system/mod.rs:19-206 synchronize_system, rfw/src/lib.rs:411-430 render_system incl. resize), error reporting,
threading, and scene edits — each checked against the oracle driven through the same calls."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pair(kind, w, h, a=0, b=0, seed=1, mpl=2):
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build(kind, a, b, 0.0, seed)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=mpl)
    orc = Oracle(w, h, threads=4, max_path_length=mpl)
    scene.sync(be)
    scene.mark_all_changed()
    scene.sync(orc)
    return scene, be, orc


def same(be, orc):
    return np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))


def test_resize_restarts_accumulation_and_matches_fresh_instance():
    from oracle.bindings import Oracle
    scene, be, orc = pair("cornell", 48, 32)
    be.render(scene.view(48, 32))
    be.resize((80, 56), 1.0)                      # render_system handles a resize before get_view (rfw/src/lib.rs:415-421)
    scene.set_aspect(80 / 56)
    view = scene.view(80, 56)
    be.render(view)
    assert be.frame_stats()["sample_count"] == 1 and be.accumulator().shape == (56, 80, 4)
    orc2 = Oracle(80, 56, threads=4, max_path_length=2)
    scene.mark_all_changed()
    scene.sync(orc2)
    orc2.render(view)
    assert same(be, orc2)


def test_material_and_light_edits_take_effect():
    from rfw_rs_amd import into_device_material
    scene, be, orc = pair("cornell", 64, 48)
    view = scene.view(64, 48)
    be.render(view); orc.render(view)
    before = be.accumulator().copy()
    assert same(be, orc)
    mats = [into_device_material([0.2, 0.3, 0.9, 1.0], [0, 0, 0.5, 0.9] + [0] * 4 + [0, 1, 0, 1] + [0] * 4),
            into_device_material([0.9, 0.1, 0.1, 1.0], [1.0, 0, 0.5, 0.2] + [0] * 4 + [0, 1, 0, 1] + [0] * 4),
            into_device_material([0.1, 0.8, 0.1, 1.0], [0, 0, 0.5, 1.0] + [0] * 4 + [1.0, 0.5, 0, 1] + [0] * 4),
            into_device_material([12.0, 12.0, 9.0, 1.0], [0] * 16)]
    be.set_materials(mats)
    be.synchronize()
    arr = (type(mats[0]) * 4)(*mats)
    orc._l.orc_set_materials(orc._h, arr, 4, None)
    orc.reset()
    be.render(view); orc.render(view)
    assert be.frame_stats()["sample_count"] == 1       # synchronize() with changes restarted the accumulation
    assert same(be, orc) and not np.array_equal(before, be.accumulator())


def test_unload_mesh_removes_its_instances():
    scene, be, orc = pair("gallery", 64, 48, seed=3)
    view = scene.view(64, 48)
    be.render(view); orc.render(view)
    assert same(be, orc) and be.scene_stats()["instances"] == 2
    be.unload_3d_meshes([1])                             # the icosphere mesh and its instance list
    be.synchronize()
    ids = (C.c_uint32 * 1)(1)
    orc._l.orc_unload_3d_meshes(orc._h, ids, 1)
    orc._l.orc_synchronize(orc._h)
    orc.reset()
    be.render(view); orc.render(view)
    assert be.scene_stats()["instances"] == 1 and same(be, orc)


def test_errors_are_reported_not_thrown():
    from rfw_rs_amd import BackendError, HipBackend, hip_lib
    lib = hip_lib()
    with pytest.raises(BackendError):
        HipBackend.init(0, 16)
    with pytest.raises(BackendError):
        HipBackend.init(16, 16, 1.0, rank=2, world=2)
    be = HipBackend.init(16, 16)
    assert lib.rfw_hip_set_3d_mesh(be._h, 0, None) < 0 and b"null" in lib.rfw_hip_last_error(be._h)
    assert lib.rfw_hip_render(be._h, None, None, 0) < 0
    assert lib.rfw_hip_set_option(be._h, b"no_such_option", 1.0) < 0 and b"unknown" in lib.rfw_hip_last_error(be._h)
    assert lib.rfw_hip_read_framebuffer(be._h, None, 4) < 0
    assert lib.rfw_hip_set_2d_mesh(be._h, 0, None, 0, -1) == 0 and lib.rfw_hip_set_2d_instances(be._h, 0, None, 0) == 0  # accepted, ignored
    assert lib.rfw_hip_set_3d_mesh(None, 0, None) < 0      # null instance
    o = np.zeros((4, 3), np.float32)
    with pytest.raises(BackendError):
        be.intersect(o, o + 1)                             # scene not synchronized yet


def test_calls_from_other_threads():
    """bevy runs synchronize_system / render_system on arbitrary workers (rfw/src/system/mod.rs:16-17): the library must not
    depend on the calling thread (it sets the device on entry and serialises on one mutex)."""
    scene, be, orc = pair("cornell", 48, 48)
    view = scene.view(48, 48)
    errs = []

    def work():
        try:
            for _ in range(3):
                be.render(view)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work) for _ in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and be.frame_stats()["sample_count"] == 9
    for _ in range(9):
        orc.render(view)
    assert same(be, orc)


def test_bandwidth_probe_reports_a_plausible_hbm_rate():
    from rfw_rs_amd import HipBackend
    be = HipBackend.init(64, 64, 1.0)
    gbs = be.bandwidth_probe(1 << 28, 10)
    assert 500.0 < gbs < 9000.0, gbs        # MI355X HBM3E: ~8 TB/s peak; a copy reaches a large fraction of it
    with pytest.raises(Exception):
        be.bandwidth_probe(0, 1)
    be.close()


def test_issue_probe_reports_plausible_vector_issue_rates():
    """rfw_hip_issue_probe: what bench.py holds the (issue-bound) trace kernels against beside the data-sheet peak.  The node test's instruction
    mix — conversions, min / max, compares, packed FMAs — issues slower than v_fma_f32 alone, and neither exceeds one wave64 instruction per
    2 cycles per SIMD at 2.4 GHz."""
    from rfw_rs_amd import BackendError, HipBackend
    be = HipBackend.init(64, 64, 1.0)
    fma, mix, packet = be.issue_probe(0, 2000), be.issue_probe(1, 2000), be.issue_probe(2, 2000)
    assert 300.0 < mix < fma < 1300.0, (fma, mix)
    # the packet kernel's node step: plain FMAs with a scalar operand plus min / max / compare, and 27 scalar instructions per 44 vector ones
    # taking issue slots beside them — its VECTOR rate lies below the FMA-only rate
    assert 200.0 < packet < fma, (fma, packet)
    with pytest.raises(BackendError):
        be.issue_probe(3)
    with pytest.raises(BackendError):
        be.issue_probe(0, 0)
    be.close()


@pytest.mark.parametrize("builder", [2, 3])
@pytest.mark.parametrize("splits", [False, True])
def test_device_built_trees_are_structurally_valid(builder, splits):
    """Reads the device builders' f32 BVH4 back and checks it the way bvh_host.cpp's validate_bvh4 checks the host builder's: every
    primitive in exactly one leaf, every child box inside its parent's and around its triangles' padded boxes.  With spatial splits the
    primitives are REFERENCES — the caller's triangles, then the duplicates of split ones — and a reference of a split triangle sits in a
    leaf box that holds its PART of the triangle: inside the triangle's box, not around it."""
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("atrium", 40000, 0, 0.0, 0xC0FFEE) if splits else Scene().build("soup", 6000, 1, 0.0, 5)
    be = HipBackend.init(32, 32, 1.0, builder=builder)
    be.set_option("spatial_splits", 8e-5 if splits else 0.0)
    scene.sync(be)
    st = be.scene_stats()
    n_tri = st["triangles"]
    assert (st["split_references"] > 0) == splits
    n = n_tri + st["split_references"]
    tris = be.debug_read("triangles", n * 176).view(np.float32).reshape(n, 44)
    v = tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(n, 3, 3)
    lo, hi = v.min(axis=1), v.max(axis=1)
    e = np.float32(1e-4) + np.float32(4e-6) * np.maximum(np.abs(lo), np.abs(hi))
    lo, hi = lo - e, hi + e                                        # the padding of launch_triangle_boxes
    nodes = be.debug_read("blas_raw", n * 128).view(np.float32).reshape(-1, 32)
    order = be.debug_read("blas_order", n * 4).view(np.uint32)
    child = nodes[:, 24:28].view(np.uint32)
    seen = np.zeros(n, np.int32)
    stack = [(0, np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32))]
    visited = 0
    n_parts = 0
    while stack:
        i, plo, phi = stack.pop()
        visited += 1
        for k in range(4):
            c = int(child[i, k])
            if c == 0xFFFFFFFF:
                continue
            clo = nodes[i, [0 + k, 8 + k, 16 + k]]
            chi = nodes[i, [4 + k, 12 + k, 20 + k]]
            assert np.all(clo >= plo) and np.all(chi <= phi), (i, k)
            if c & 0x80000000:
                first, count = c & 0x07FFFFFF, ((c >> 27) & 15) + 1
                ids = order[first:first + count]
                seen[ids] += 1
                inside = np.all(lo[ids] >= clo, axis=1) & np.all(hi[ids] <= chi, axis=1)
                if not splits:
                    assert inside.all(), (i, k)
                else:  # a part of a split triangle: the leaf box lies inside the triangle's box (and is not empty)
                    part = ~inside
                    n_parts += int(part.sum())
                    assert np.all((clo >= lo[ids[part]] - 1e-3) & (chi <= hi[ids[part]] + 1e-3) & (chi >= clo)), (i, k)
            else:
                if builder == 3:
                    assert c > i                                   # the SAH builder numbers children after their parents
                stack.append((c, clo, chi))
    assert np.all(seen == 1)
    assert visited <= be.scene_stats()["blas_nodes"]
    if splits:
        # the duplicates are copies of their originals (same vertices), and only references of split triangles sit in part boxes
        assert 1 <= n_parts <= 2 * st["split_references"] + 64
    be.close()


def test_fence_free_tlas_fit_survives_ten_thousand_rebuilds():
    """The per-frame TLAS builder hands a subtree's box from the first child to arrive at a node to the second WITHOUT fences (csrc/lbvh.hip,
    k_fit: write-through stores, s_waitcnt, a relaxed counter, sc1 loads).  10 000 rebuilds of a tree over 10 000 jittered boxes — every tree
    checked on the device, exactly: each child box equals the union of what lies below it, each box sits in one leaf — while a second process
    keeps the device busy, so that the wavefronts of a launch do NOT all start together.  RFW_LBVH_FENCED=1 is the known-good fallback."""
    import subprocess, sys, os
    from rfw_rs_amd import HipBackend
    hog = subprocess.Popen([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r)\nfrom rfw_rs_amd import HipBackend\nbe = HipBackend.init(64, 64, 1.0)\n"
                            "import time\nt = time.time()\nwhile time.time() - t < 60: be.bandwidth_probe(1 << 28, 20)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))])
    try:
        be = HipBackend.init(32, 32, 1.0)
        total_checked = 0
        for k in range(10):
            errors, checked = be.lbvh_stress(10_000, 1_000, seed=1 + k)
            assert errors == 0, (k, errors, checked)
            total_checked += checked
        assert total_checked > 10_000 * 10_000  # every primitive's leaf box and every interior child box, every rebuild
        # a few large trees too (1 M boxes: the BLAS-sized case of builder = DEVICE_LBVH)
        errors, checked = be.lbvh_stress(1_000_000, 5, seed=99)
        assert errors == 0 and checked > 5_000_000, (errors, checked)
        be.close()
    finally:
        hog.kill()
        hog.wait()


def test_depth_test_returns_the_closest_hit_and_a_node_count():
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("soup", 4000, 3, 0.0, 9)
    be = HipBackend.init(32, 32, 1.0)
    orc = Oracle(32, 32, threads=2)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    rng = np.random.default_rng(4)
    o = rng.uniform(-4, 4, (3000, 3)).astype(np.float32)
    d = rng.normal(size=(3000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits, depth = be.depth_test(o, d)
    ref = orc.intersect(o, d, brute=True)
    assert np.array_equal(hits["tri"], ref["tri"]) and np.array_equal(hits["t"][ref["inst"] >= 0].view(np.uint32), ref["t"][ref["inst"] >= 0].view(np.uint32))
    assert np.all(depth >= 1) and depth.max() < 2000              # every ray visits at least the TLAS root
    assert depth[hits["inst"] >= 0].mean() > depth[hits["inst"] < 0].mean() * 0.5
    assert np.array_equal(be.intersect(o, d)["tri"], hits["tri"])  # the same query without the counter
    be.close()


def test_download_frame_is_the_frame_and_does_not_stall_the_slots():
    """rfw_hip_download_frame queues a copy of the latest frame into pinned host memory behind its kernels; several can be outstanding
    (one per frame slot); after wait_downloads each buffer holds exactly what read_framebuffer / read_accumulator return."""
    from rfw_rs_amd import HipBackend, Scene
    w, h = 96, 64
    scene = Scene().build("soup", 1500, 3, 0.0, 2)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2, frames_in_flight=3)
    scene.sync(be)
    views, bufs, want = [], [], []
    for k in range(5):
        scene.set_camera([0.2 * k - 0.4, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
        bufs.append((be.host_frame(), be.host_frame()))
    for v, (fb, acc) in zip(views, bufs):
        be.render(v)
        be.download_frame(fb)
        be.download_frame(acc, accumulator=True)
    be.wait_downloads()
    ref = HipBackend.init(w, h, 1.0, max_path_length=2)
    scene.mark_all_changed(); scene.sync(ref)
    for v, (fb, acc) in zip(views, bufs):
        ref.reset_accumulation(); ref.render(v)
        assert np.array_equal(fb.view(np.uint32), ref.framebuffer().view(np.uint32))
        assert np.array_equal(acc.view(np.uint32), ref.accumulator().view(np.uint32))
    with pytest.raises(Exception):
        be.download_frame(np.zeros((4, 4, 4), np.float32))
    # the presented frame: gpu-rt's Bgra8UnormSrgb swap chain image of the same frame (B, G, R, A bytes)
    steps = be.srgb_steps()
    e = (np.arange(255) + 0.5) / 255.0
    exact = np.where(e <= 0.04045, e / 12.92, ((e + 0.055) / 1.055) ** 2.4)
    assert np.all(np.diff(steps) > 0) and np.all(np.abs(steps.astype(np.float64) - exact) <= np.spacing(steps).astype(np.float64))
    for lin, byte in ((0.0, 0), (1.0, 255), (7.0, 255), (0.5, 188), (0.18, 118), (0.0031308 / 2, 5), (-1.0, 0)):
        assert np.searchsorted(steps, np.float32(lin), side="right") == byte, lin
    pres = be.host_frame(presented=True)
    be.render(views[2]); be.download_frame(pres); be.wait_downloads()
    ref.reset_accumulation(); ref.render(views[2])
    fb = ref.framebuffer()
    want = np.searchsorted(steps, fb[..., :3], side="right").astype(np.uint8)
    assert np.array_equal(pres[..., 0], want[..., 2]) and np.array_equal(pres[..., 1], want[..., 1]) and np.array_equal(pres[..., 2], want[..., 0])
    assert np.all(pres[..., 3] == 255) and pres[..., :3].max() > 40
    be.close(); ref.close()


def test_cpp_example_animated_runs_the_reference_frame_loop(tmp_path):
    """host/example_animated.cpp: the reference's examples/animated as a compiled host program (no Python in the loop): a glTF scene written
    by the exporter, a grid of bouncing instances, synchronize_system + render_system every frame, every frame presented to host memory."""
    import os
    import subprocess
    from conftest import ROOT
    from rfw_rs_amd import Scene
    exe = os.path.join(ROOT, "rfw-rs_amd", "host", "example_animated")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C rfw-rs_amd/host example_animated)"
    glb = Scene().build("atrium", 30000, 0, 0.0, 3).save_glb(str(tmp_path / "atrium.glb"))
    out = tmp_path / "last.ppm"
    r = subprocess.run([exe, "--gltf", glb, "--frames", "40", "--size", "320x200", "--spheres", "20x20", "--out", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "40 frames of 320x200" in r.stdout and "400 animated instances" in r.stdout
    raw = out.read_bytes()
    head = b"P6\n320 200\n255\n"
    assert raw.startswith(head)
    img = np.frombuffer(raw[len(head):], np.uint8).reshape(200, 320, 3)
    assert img.mean() > 8 and img.std() > 8          # a lit, structured image
    bad = subprocess.run([exe, "--gltf", str(tmp_path / "missing.glb")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert bad.returncode == 1 and "missing.glb" in bad.stderr
    # the reference's two animated actors (main.rs:79-103) and its animation timer system: two graphs of an animated, skinned glTF document,
    # Scene::set_animations_time every frame -> skinning and refit on the device; the image differs from the run without them
    from gltf_util import write_animated_gltf
    actor = str(write_animated_gltf(tmp_path))
    out2 = tmp_path / "actors.ppm"
    r2 = subprocess.run([exe, "--gltf", glb, "--actor", actor, "--actor", actor, "--frames", "40", "--size", "320x200", "--spheres", "20x20", "--out", str(out2)],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    img2 = np.frombuffer(out2.read_bytes()[len(head):], np.uint8).reshape(200, 320, 3)
    assert (img2 != img).mean() > 0.01


def test_batch_and_accumulator_corner_cases():
    """Error paths of rfw_hip_render_batch, the on-demand accumulator frame before the first frame and after several samples, and a
    batch on an instance created without max_batch."""
    from rfw_rs_amd import BackendError, HipBackend
    scene, be, orc = pair("cornell", 40, 24)
    assert not be.accumulator().any() and not be.framebuffer().any()        # nothing rendered yet: zeros, not garbage
    v = scene.view(40, 24)
    with pytest.raises(BackendError):
        be.render_batch([v, v])                                              # options.max_batch was not set
    for _ in range(3):
        be.render(v); orc.render(v)
    assert same(be, orc)                                                     # accumulator de-tiled on demand, three samples in it
    acc = be.host_frame()
    be.download_frame(acc, accumulator=True); be.wait_downloads()
    assert np.array_equal(acc.view(np.uint32), orc.accumulator().view(np.uint32))
    be.close()
    b2 = HipBackend.init(40, 24, 1.0, max_path_length=2, max_batch=3)
    scene.mark_all_changed(); scene.sync(b2)
    scene.set_camera([0, 0, -3.4], [0, 0, 1], fov=60.0, aspect=40 / 24)
    wide = scene.view(40, 24)
    with pytest.raises(BackendError):
        b2.render_batch([v, wide])                                           # the views of a batch must share one spread angle
    with pytest.raises(BackendError):
        b2.render_batch([v] * 4)                                             # more than max_batch
    with pytest.raises(BackendError):
        HipBackend.init(40, 24, 1.0, streams=2, max_batch=2)                 # sub-streams and batches exclude each other
    b2.render_batch([v, v, v])
    orc.reset(); orc.render(v)
    for f in range(3):
        assert np.array_equal(b2.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32))
    b2.resize((24, 16), 1.0)                                                 # resize drops the frames: zeros again
    assert not b2.accumulator().any()
    b2.close()


def test_library_owned_collective_with_a_one_rank_communicator():
    """the all-gather inside the library (rfw_hip_comm_init: librccl opened at run time, ncclAllGather on the instance's
    stream).  One GPU allows a one-rank communicator: render() then packs, gathers (to itself) and de-tiles the frame on its own, and the
    result must be the frame of an instance without a communicator, for single frames, accumulation and batches."""
    from rfw_rs_amd import BackendError, HipBackend, Scene
    w, h = 200, 136
    scene = Scene().build("soup", 1500, 5, 0.0, 4)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    plain = HipBackend.init(w, h, 1.0, max_path_length=3, max_batch=3)
    scene.sync(plain)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, max_batch=3)
    scene.mark_all_changed(); scene.sync(be)
    uid = HipBackend.comm_unique_id()
    assert len(uid) == 128
    with pytest.raises(BackendError):
        be.comm_init(uid, 1, 2)                       # not the shard this instance was created with
    be.comm_init(uid, 0, 1)
    with pytest.raises(BackendError):
        be.comm_init(uid, 0, 1)                       # already has one
    for _ in range(2):
        plain.render(view); be.render(view)
    assert be.frame_stats()["sample_count"] == 2
    assert np.array_equal(be.framebuffer().view(np.uint32), plain.framebuffer().view(np.uint32))
    a, b = be.accumulator(), plain.accumulator()
    assert np.array_equal(a[..., :3].view(np.uint32), b[..., :3].view(np.uint32))   # the gathered slabs carry RGB (alpha is never written)
    views = []
    for i in range(3):
        scene.set_camera([0.3 * i - 0.3, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    plain.render_batch(views); be.render_batch(views)
    for f in range(3):
        assert np.array_equal(be.framebuffer_at(f).view(np.uint32), plain.framebuffer_at(f).view(np.uint32)), f
    be.resize((96, 64)); plain.resize((96, 64))
    scene.set_aspect(96 / 64)
    v2 = scene.view(96, 64)
    be.render(v2); plain.render(v2)
    assert np.array_equal(be.framebuffer().view(np.uint32), plain.framebuffer().view(np.uint32))
    be.comm_destroy()
    be.reset_accumulation(); plain.reset_accumulation()
    be.render(v2); plain.render(v2)
    assert np.array_equal(be.framebuffer().view(np.uint32), plain.framebuffer().view(np.uint32))
    be.close(); plain.close()


def test_frame_slots_share_one_communicator():
    """VERDICT r02 #6d: frames in flight of a sharded frame without a scene copy per frame.  One instance, three frame slots, ONE
    communicator: every slot gathers into buffers of its own on its own stream, the collectives are chained on the device.  Frames rendered
    round-robin over the slots equal the frames of a plain instance, for every gather format."""
    from rfw_rs_amd import HipBackend, Scene
    w, h = 200, 136
    scene = Scene().build("soup", 1500, 5, 0.0, 4)
    scene.set_aspect(w / h)
    views = []
    for i in range(7):
        scene.set_camera([0.2 * i - 0.6, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    plain = HipBackend.init(w, h, 1.0, max_path_length=3)
    scene.sync(plain)
    want = []
    for v in views:
        plain.reset_accumulation(); plain.render(v)
        pres = plain.host_frame(presented=True)
        plain.download_frame(pres); plain.wait_downloads()
        want.append((plain.framebuffer().copy(), pres.copy()))
        plain.free_host_frame(pres)
    for fmt in (0, 1, 2):
        be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=3)
        scene.mark_all_changed(); scene.sync(be)
        be.set_option("gather_format", fmt)
        be.comm_init(HipBackend.comm_unique_id(), 0, 1)
        host = [be.host_frame(presented=(fmt == 2)) for _ in views]
        for v, dst in zip(views, host):     # nothing waits between the frames: three are in flight
            be.render(v)
            be.download_frame(dst)
        be.wait_downloads()
        for i, dst in enumerate(host):
            if fmt == 0:
                assert np.array_equal(dst.view(np.uint32), want[i][0].view(np.uint32)), (fmt, i)
            elif fmt == 1:
                assert np.array_equal(dst[..., :3].astype(np.float16).view(np.uint16), want[i][0][..., :3].astype(np.float16).view(np.uint16)), (fmt, i)
            else:
                assert np.array_equal(dst, want[i][1]), (fmt, i)
        be.comm_destroy()
        be.close()
    plain.close()


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_exchange_by_peer_stores(fmt, present_rank=0):
    """VERDICT r02 #6c / SURVEY §8e's alternative to the all-gather: every rank packs its tiles straight into the destinations' receive buffers
    and raises their arrival flags; a destination polls its own flags, de-tiles, and returns credits.  One GPU holds one rank per instance
    here (three ranks of one process: peers are reached through their addresses, the multi-process mapping through hipIpcOpenMemHandle is
    what tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu_p2p runs), with two frame slots each so that the credit of frame k
    gates frame k + 2.  The presenting rank ends up with the single-GPU frame; a rank that only sends has no frame.  (Every rank a
    destination, present_rank = -1, needs streams that wait for LATER launches of other streams: on one device that deadlocks as soon as
    two of them share a hardware queue, so that case runs with one process per rank in the bench test.)"""
    from rfw_rs_amd import BackendError, HipBackend, Scene
    w, h = 200, 136
    scene = Scene().build("soup", 1500, 5, 0.0, 4)
    scene.set_aspect(w / h)
    views = []
    for i in range(5):
        scene.set_camera([0.2 * i - 0.4, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    plain = HipBackend.init(w, h, 1.0, max_path_length=3)
    scene.sync(plain)
    world = 3
    ranks = []
    for r in range(world):
        be = HipBackend.init(w, h, 1.0, max_path_length=3, rank=r, world=world, tile_size=32, frames_in_flight=2)
        be.set_option("gather_format", fmt)
        be.set_option("present_rank", present_rank)
        be.set_option("p2p_timeout_ms", 3000)
        scene.mark_all_changed(); scene.sync(be)
        ranks.append(be)
    handles = [be.p2p_export() for be in ranks]
    assert all(len(hd) == 256 for hd in handles)
    with pytest.raises(BackendError):
        ranks[0].p2p_connect(handles[::-1])           # not in rank order
    handles = [be.p2p_export() for be in ranks]       # (the failed connect released rank 0's buffers)
    for be in ranks:
        be.p2p_connect(handles)
    with pytest.raises(BackendError):
        ranks[0].resize((96, 64))                     # the buffers are sized for the frame
    with pytest.raises(BackendError):
        ranks[0].comm_init(HipBackend.comm_unique_id(), 0, world)
    dests = range(world) if present_rank < 0 else [present_rank]
    host = {r: [ranks[r].host_frame(presented=(fmt == 2)) for _ in views] for r in dests}
    for i, v in enumerate(views):                     # nothing waits on the host: the streams wait for each other's flags
        for r in reversed(range(world)):              # (senders first: with one presenting rank nobody ever waits for a later launch)
            ranks[r].render(v)
            if r in host:
                ranks[r].download_frame(host[r][i], accumulator=False)
    for r in dests:
        ranks[r].wait_downloads()
    for i, v in enumerate(views):
        plain.reset_accumulation(); plain.render(v)
        if fmt == 2:
            want = plain.host_frame(presented=True)
            plain.download_frame(want); plain.wait_downloads()
        else:
            want = plain.framebuffer()
        for r in dests:
            if fmt == 1:
                assert np.array_equal(host[r][i][..., :3].astype(np.float16).view(np.uint16), want[..., :3].astype(np.float16).view(np.uint16)), (r, i)
            else:
                assert np.array_equal(host[r][i].view(np.uint32), want.view(np.uint32)), (r, i)
    if present_rank >= 0:
        with pytest.raises(BackendError):
            ranks[1].framebuffer()                    # rank 1 only sent its tiles
    for be in ranks:
        be.device_synchronize()
    for be in ranks:
        be.p2p_disconnect()
    ranks[0].render(views[0])                         # a sharded instance without an exchange again
    for be in ranks + [plain]:
        be.close()


def test_exchange_by_peer_stores_times_out_instead_of_hanging():
    """A peer that never renders: the wait gives up after p2p_timeout_ms and the next read fails."""
    from rfw_rs_amd import BackendError, HipBackend, Scene
    w, h = 96, 64
    scene = Scene().build("cornell")
    scene.set_aspect(w / h)
    ranks = []
    for r in range(2):
        be = HipBackend.init(w, h, 1.0, rank=r, world=2, tile_size=32)
        be.set_option("p2p_timeout_ms", 200)
        scene.mark_all_changed(); scene.sync(be)
        ranks.append(be)
    handles = [be.p2p_export() for be in ranks]
    for be in ranks:
        be.p2p_connect(handles)
    ranks[0].render(scene.view(w, h))                 # rank 1 never sends
    ranks[0].device_synchronize()
    with pytest.raises(BackendError):
        ranks[0].framebuffer()
    for be in ranks:
        be.p2p_disconnect(); be.close()


def test_packet_queries_are_four_single_ray_queries():
    """TIntersector::intersect4 / occludes4 (intersector.rs:129-166) in rtbvh's SoA packet layout."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("soup", 1200, 3, 0.0, 2)
    be = HipBackend.init(32, 32)
    orc = Oracle(32, 32)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    rng = np.random.default_rng(0)
    for _ in range(20):
        o = rng.uniform(-3, 3, (4, 3)).astype(np.float32)
        d = rng.normal(size=(4, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tmin = rng.uniform(1e-4, 0.5, 4).astype(np.float32)
        tmax = rng.uniform(1.0, 30.0, 4).astype(np.float32)
        ii, pp, t = be.intersect4(o, d, tmin, tmax)
        occ = be.occludes4(o, d, tmin, tmax)
        for k in range(4):
            r = orc.intersect(o[k:k + 1], d[k:k + 1], t_min=float(tmin[k]), t_max=float(tmax[k]), brute=True)
            assert ii[k] == r["inst"][0] and pp[k] == r["tri"][0]
            assert t[k] == (r["t"][0] if r["inst"][0] >= 0 else tmax[k])
            assert bool(occ[k]) == bool(orc.occludes(o[k:k + 1], d[k:k + 1], tmax[k:k + 1], t_min=float(tmin[k]), brute=True)[0])
    be.close()


def test_degenerate_rays_with_an_infinite_interval_do_not_fault():
    """Regression (found by tools/probes/wave_model.py): origin far outside the scene, a direction of exact zeros (1 / d = +inf) and
    t_max = +inf gave child boxes an entry distance of +inf that still passed `entry <= t`, such children tied with the empty slots'
    +inf keys in the ordering network, an empty slot's reference was followed and decoded as a leaf beyond the triangle array: a GPU
    memory fault through rfw_hip_occludes on the 1 M-triangle scene.  The search interval now ends at a finite distance."""
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("atrium", 1048576, 0, 0.0, 0xC0FFEE)
    scene.set_aspect(16 / 9)
    be = HipBackend.init(640, 360, 1.0, max_path_length=1)
    scene.sync(be)
    n = 60000
    rng = np.random.default_rng(2)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    far = (d * np.float32(1e26)).astype(np.float32)
    zero = np.zeros((n, 3), np.float32)
    mixed = np.where(rng.random((n, 3)) < 0.5, zero, -zero).astype(np.float32)
    inf = np.full(n, np.inf, np.float32)
    for o, dd in ((far, zero), (far, mixed), (far, -zero), (d, zero), (far, d)):
        assert not be.occludes(o, dd, inf).any()
        occ, depth = be.occludes_depth(o, dd, inf)
        assert not occ.any() and depth.max() <= 2
        hits = be.intersect(o, dd, t_max=float("inf"))
        assert (hits["inst"] < 0).all() and np.isinf(hits["t"]).all()          # a miss reports the caller's t_max
    # ordinary rays with t_max = +inf behave as with any large finite bound
    o = rng.uniform(-10, 10, (n, 3)).astype(np.float32); o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.3
    a, b = be.intersect(o, d, t_max=float("inf")), be.intersect(o, d, t_max=1e30)
    assert np.array_equal(a["inst"], b["inst"]) and np.array_equal(a["tri"], b["tri"]) and (a["inst"] >= 0).mean() > 0.5
    assert np.array_equal(be.occludes(o, d, inf), be.occludes(o, d, np.full(n, 1e30, np.float32)))
    be.close()


def test_resent_meshes_upload_heads_first_and_give_the_same_image():
    """A full device build sends the 48-B heads of the records first and builds the trees while the 176-B records follow on a second stream
    (DESIGN.md §4, round 4).  It only does so for meshes the caller RE-sends (their host copies are registered with the runtime on the first
    re-send).  Whatever path a build takes, the scene on the device is the same: frames before and after are bit-identical, and a changed
    mesh shows exactly as on a backend that was given the final scene once."""
    from rfw_rs_amd import HipBackend, Scene
    w, h = 256, 144
    scene = Scene().build("atrium", 700000, 1, 0.0, 11)     # one large mesh + 64 spheres of 5120 triangles (0.9 MB each: registered too)
    scene.set_aspect(w / h)
    view = scene.view(w, h)

    def frame(be):
        be.reset_accumulation(); be.render(view); be.device_synchronize()
        return be.framebuffer().copy()

    def counters(be):
        return [int(x) for x in be.debug_read("build_counters", 16).view(np.uint32)]

    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    scene.sync(be)
    first = frame(be)
    assert counters(be)[2:] == [0, 0]                       # a scene sent once: nothing registered, the plain upload
    scene.mark_all_changed(); scene.sync(be)                # first re-send: the host copies get registered -> this build already goes heads first
    c = counters(be)
    assert c[0] == 2 and c[2] == 1 and c[3] == scene.counts()["meshes"] >= 2, c
    assert np.array_equal(frame(be).view(np.uint32), first.view(np.uint32))
    scene.mark_all_changed(); scene.sync(be)
    assert counters(be)[2] == 2
    assert np.array_equal(frame(be).view(np.uint32), first.view(np.uint32))
    # new content through the same path: two spheres replaced, everything re-sent
    scene.replace_mesh_with_sphere(3, 2, 77)
    scene.replace_mesh_with_sphere(9, 4, 78)
    scene.mark_all_changed(); scene.sync(be)
    assert counters(be)[2] == 3
    changed = frame(be)
    assert not np.array_equal(changed.view(np.uint32), first.view(np.uint32))
    ref = HipBackend.init(w, h, 1.0, max_path_length=2)
    scene.mark_all_changed(); scene.sync(ref)
    assert np.array_equal(changed.view(np.uint32), frame(ref).view(np.uint32))
    # ... and one mesh alone afterwards (the incremental path, from the registered copy)
    scene.replace_mesh_with_sphere(5, 3, 79)
    scene.sync(be); scene.mark_all_changed(); scene.sync(ref)
    assert counters(be)[1] >= 1
    assert np.array_equal(frame(be).view(np.uint32), frame(ref).view(np.uint32))
    ref.close()
    be.close()
    # a scene of ONE large mesh that is re-sent: the incremental path, heads first from the second re-send on
    scene = Scene().build("atrium", 120000, 0, 0.0, 5)
    assert scene.counts()["meshes"] == 1                    # (everything changed = one mesh changed: the incremental path)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    scene.sync(be)
    first = frame(be)
    before = counters(be)
    for k in range(3):
        scene.mark_all_changed()
        scene.sync(be)
        assert np.array_equal(frame(be).view(np.uint32), first.view(np.uint32)), k
    after = counters(be)
    assert after[1] == before[1] + 3 and after[2] >= before[2] + 2, (before, after)   # (the first re-send registers the copy and already goes heads first)
    be.close()


def test_backends_one_after_the_other_in_one_process():
    """Every builder x (no frame slots, three) on small scenes, one backend after the other in ONE process, two frames back to back against the
    oracle.  A later backend gets device memory an earlier one has used: what a fresh process hands out zeroed is not zero here.  Found by
    tests/soak_gpu.py in round 4: the host builder's per-slot TLAS was uploaded on the owner's stream and expanded into its octant copies on the
    slot's — the expansion could read nodes that had not arrived, and only a recycled allocation showed it."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    rng = np.random.default_rng(7)
    w, h = 96, 70
    for rep in range(2):
        for builder in (0, 1, 2, 3):
            for fif in (3, 0):
                tris, inst, seed = int(rng.integers(300, 5000)), int(rng.integers(1, 20)), int(rng.integers(1, 1 << 30))
                scene = Scene().build("soup", tris, inst, 0.0, seed)
                scene.set_aspect(w / h)
                be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder, frames_in_flight=fif)
                orc = Oracle(w, h, threads=8, max_path_length=3)
                scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
                view = scene.view(w, h)
                for _ in range(2):
                    be.render(view); orc.render(view)
                assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), (rep, builder, fif, tris, inst)
                be.close()


@pytest.mark.parametrize("n_inst", [2, 3, 17, 1000, 10000])
def test_fused_tlas_build_equals_the_chain(n_inst):
    """up to 16 384 instances the per-frame TLAS is built by ONE workgroup (csrc/lbvh.hip, k_tlas_fused) and finished by one more
    launch (k_tlas_finish), instead of the chain of 23 launches (option tlas_fused = 0 keeps the chain).  Same arithmetic, so the same tree: the
    4-wide nodes before and after quantisation, every per-octant copy and the leaf order are compared byte for byte, over several frames of an
    animation, and both equal the oracle's image."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 96, 64
    if n_inst >= 1000:
        side = int(round(n_inst ** 0.5))
        scene = Scene().build("cornell")
        scene.build("spheres", side, side, 0.28)          # + side x side animated icosphere instances
    else:
        scene = Scene().build("soup", 900, n_inst, 0.0, 21)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    fused, chain = HipBackend.init(w, h, 1.0), HipBackend.init(w, h, 1.0)
    fused.set_option("tlas_fused", 1)    # (the default, 2, takes the one-workgroup build only where the instance has frame slots)
    chain.set_option("tlas_fused", 0)
    orc = Oracle(w, h, threads=8)
    for frame in range(3):
        if n_inst >= 1000:
            scene.animate(frame / 7.0)
        for be in (fused, chain):
            scene.mark_all_changed()
            scene.sync(be)
        scene.mark_all_changed(); scene.sync(orc)
        nf, nc = fused.scene_stats()["tlas_nodes"], chain.scene_stats()["tlas_nodes"]
        assert nf == nc and nf >= 1
        n_valid = fused.scene_stats()["instances"]
        for what, item in (("tlas_raw", 128), ("tlas_nodes", 64)):
            a, b = fused.debug_read(what, nf * item), chain.debug_read(what, nf * item)
            assert len(a) == nf * item and np.array_equal(a, b), (frame, what, np.argwhere(a != b)[:4])
        a, b = fused.debug_read("tlas_prims", 4 * n_valid), chain.debug_read("tlas_prims", 4 * n_valid)
        assert len(a) > 0 and np.array_equal(a, b), (frame, "tlas_prims")
        fo, co = fused.debug_read("tlas_oct", 1 << 30), chain.debug_read("tlas_oct", 1 << 30)
        fo, co = fo.reshape(8, -1), co.reshape(8, -1)          # (the strides may differ: each follows its own allocation)
        assert np.array_equal(fo[:, : nf * 64], co[:, : nf * 64]), (frame, "tlas_oct")
        orc.reset()
        for be in (fused, chain):
            be.reset_accumulation()
            be.render(view)
        orc.render(view)
        assert np.array_equal(fused.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), frame
        assert np.array_equal(chain.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), frame
    cf = fused.debug_read("build_counters", 20).view(np.uint32)
    cc = chain.debug_read("build_counters", 20).view(np.uint32)
    assert cf[4] >= 3 and cc[4] == 0, (cf, cc)          # the fused path did run — and only where it was asked to
    fused.close(); chain.close()
    # the default: fused where frames overlap (frame slots), the chain one frame at a time
    for slots, want_fused in ((0, False), (3, True)):
        be = HipBackend.init(w, h, 1.0, frames_in_flight=slots)
        scene.mark_all_changed(); scene.sync(be)
        be.render(view); be.device_synchronize()
        used = int(be.debug_read("build_counters", 20).view(np.uint32)[4])
        assert (used > 0) == want_fused, (slots, used)
        be.close()
