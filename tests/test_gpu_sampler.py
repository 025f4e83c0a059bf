"""Round 2 additions on the device: the blue-noise sampler of the first 256 samples (tables as run-time input), several samples of
one image in one launch per stage (rfw_hip_render_samples), and the contained traversal-stack overflow."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def blue_noise_table(seed):
    # layout of gpu_rt::blue_noise::create_blue_noise_buffer(): 5 x 65536 words, each a byte value
    return np.random.default_rng(seed).integers(0, 256, 5 * 65536).astype(np.uint32)


def make(kind, w, h, a=0, b=0, seed=1, **opts):
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build(kind, a, b, 0.0, seed)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, **opts)
    scene.sync(be)
    orc = Oracle(w, h, threads=8, max_path_length=opts.get("max_path_length", 3))
    scene.mark_all_changed()
    scene.sync(orc)
    return scene, be, orc


@pytest.mark.parametrize("kind,a,b", [("cornell", 0, 0), ("soup", 1500, 5), ("gallery", 0, 0)])
def test_blue_noise_frames_match_oracle(kind, a, b):
    """ray_gen.comp:109-122, shade.comp:189-227: samples < 256 draw their 4 + 4 per bounce numbers from the tables."""
    w, h = 160, 136   # wider than one 128 x 128 blue-noise tile in both directions
    scene, be, orc = make(kind, w, h, a, b, seed=4, max_path_length=3)
    view = scene.view(w, h)
    t = blue_noise_table(5)
    be.render(view); orc.render(view)
    plain = be.accumulator().copy()
    assert np.array_equal(plain.view(np.uint32), orc.accumulator().view(np.uint32))
    be.set_blue_noise(t); orc.set_blue_noise(t); orc.reset()
    for s in range(3):
        be.render(view); orc.render(view)
        assert be.frame_stats()["sample_count"] == s + 1      # set_blue_noise restarted the image
        ga, ra = be.accumulator(), orc.accumulator()
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), s
    assert not np.array_equal(plain, be.accumulator())
    # across the switch to xorshift at sample 256 (the image so far is the same on both sides)
    be.set_option("sample_count", 254); orc.set_option("sample_count", 254)
    for s in range(4):
        be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), 254 + s
    # the primary rays themselves (pixel jitter = dimensions 0 and 1 of the sampler)
    po, pd = orc.primary_rays(view, 0)
    assert np.array_equal(be.intersect(po, pd)["tri"], orc.intersect(po, pd)["tri"])
    # clearing the tables restores the xorshift image
    be.set_blue_noise(None); orc.set_blue_noise(None); orc.reset()
    be.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), plain.view(np.uint32))
    from rfw_rs_amd import BackendError
    with pytest.raises(BackendError):
        be.set_blue_noise(t[:1000])
    with pytest.raises(BackendError):
        be.set_blue_noise(t + 256)                             # entries are bytes
    be.close()


def test_blue_noise_in_batches_and_frame_slots():
    w, h = 104, 72
    scene, be, orc = make("soup", w, h, 1200, 4, seed=8, max_path_length=3, max_batch=6, frames_in_flight=3)
    t = blue_noise_table(2)
    be.set_blue_noise(t); orc.set_blue_noise(t)
    views = []
    for i in range(5):
        scene.set_camera([0.35 * i - 0.7, 0.3 + 0.1 * i, -4.0 + 0.15 * i], [0.05 * i, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    be.render_batch(views)
    for f, v in enumerate(views):
        orc.reset(); orc.render(v)
        assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    for v in views[:4]:                                        # one render() per view over the slots
        be.render(v)
    orc.reset(); orc.render(views[3])
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.close()


@pytest.mark.parametrize("tables", [False, True])
def test_render_samples_is_k_samples_of_one_image(tables):
    """rfw_hip_render_samples(view, k): the same paths as k render() calls, one launch per stage; the accumulator is the sum of the
    per-sample images in sample order (bit for bit), i.e. the sequential accumulation up to the rounding of the partial sums."""
    from oracle.bindings import Oracle
    w, h = 120, 88
    scene, be, orc = make("soup", w, h, 1500, 5, seed=6, max_path_length=3, max_batch=4)
    view = scene.view(w, h)
    t = blue_noise_table(9) if tables else None
    be.set_blue_noise(t); orc.set_blue_noise(t); orc.reset()
    k = 4
    be.render_samples(view, k)
    assert be.frame_stats()["sample_count"] == k
    for _ in range(k):
        orc.render(view)
    ga = be.accumulator()
    assert rel_l2(ga, orc.accumulator()) <= 1e-6              # gpu-rt's sequential accumulation, to rounding
    total = np.zeros((h, w, 4), np.float32)
    for s in range(k):                                         # the image of sample s alone: a fresh accumulator, sample index s
        one = Oracle(w, h, threads=8, max_path_length=3)
        scene.mark_all_changed(); scene.sync(one)
        one.set_blue_noise(t)
        one.set_option("sample_count", s)
        one.render(view)
        total = total + one.accumulator() if s else one.accumulator().copy()
    assert np.array_equal(ga.view(np.uint32), total.view(np.uint32))
    assert np.array_equal(be.framebuffer().view(np.uint32), np.sqrt(total * np.float32(1.0) / np.float32(k)).view(np.uint32))
    # more samples of the same image keep accumulating: samples 4..6, then a plain render() as sample 7
    be.render_samples(view, 3)
    be.render(view)
    for _ in range(4):
        orc.render(view)
    assert be.frame_stats()["sample_count"] == 8
    assert rel_l2(be.accumulator(), orc.accumulator()) <= 1e-6
    # a new view starts a new image
    scene.set_camera([0.2, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
    v2 = scene.view(w, h)
    be.render_samples(v2, 2)
    orc.reset(); orc.render(v2); orc.render(v2)
    assert be.frame_stats()["sample_count"] == 2 and rel_l2(be.accumulator(), orc.accumulator()) <= 1e-6
    from rfw_rs_amd import BackendError
    with pytest.raises(BackendError):
        be.render_samples(v2, 5)                               # more than options.max_batch
    be.close()


def test_traversal_stack_overflow_is_contained_and_reported():
    """a stack deeper than LDS + spill rows must not index past the spill rows; the entry is dropped, a flag in pinned host
    memory is set and every later call reports RFW_HIP_E_STATE until synchronize() rebuilds the trees.  The test shrinks the spill
    stack to zero rows on a deep LBVH so that the path is actually taken."""
    from rfw_rs_amd import BackendError
    w, h = 64, 64
    scene, be, orc = make("soup", w, h, 40000, 1, seed=3, builder=2)   # LBVH: deeper trees than SAH
    rng = np.random.default_rng(1)
    o = rng.uniform(-4, 4, (40000, 3)).astype(np.float32)
    d = rng.normal(size=(40000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = orc.intersect(o, d)
    g = be.intersect(o, d)
    assert np.array_equal(g["tri"], ref["tri"])
    be.set_option("spill_rows", 0)
    with pytest.raises(BackendError, match="stack overflow"):
        be.intersect(o, d)                                     # completes (no fault, no out-of-bounds access) and reports
    with pytest.raises(BackendError, match="stack overflow"):
        be.render(scene.view(w, h))                            # sticky
    be.set_option("spill_rows", 48)
    scene.mark_all_changed(); scene.sync(be)                   # new trees: the flag is cleared
    g = be.intersect(o, d)
    assert np.array_equal(g["tri"], ref["tri"]) and np.array_equal(g["t"].view(np.uint32), ref["t"].view(np.uint32))
    be.render(scene.view(w, h))
    be.close()


def test_sorted_extension_rays_give_the_same_image():
    """Option sort_extension_rays: extension rays are traced in the order of a Morton key of their origin (and direction octant); the queue
    keeps its order, so counts and image are bit-identical to the oracle's — through frame slots and batches too."""
    w, h = 200, 136
    scene, be, orc = make("soup", w, h, 2500, 6, seed=13, max_path_length=4, frames_in_flight=2, max_batch=3)
    orc.set_option("max_path_length", 4)
    be.set_option("sort_extension_rays", 1)
    view = scene.view(w, h)
    for _ in range(2):
        be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    s, o = be.frame_stats(), orc.stats()
    assert s["extension_rays"] > 0
    views = []
    for i in range(3):
        scene.set_camera([0.3 * i - 0.3, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    be.render_batch(views)
    for f, v in enumerate(views):
        orc.reset(); orc.render(v)
        assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    be.close()


@pytest.mark.parametrize("run,refill", [(1, 16), (4, 1), (4, 16), (16, 48), (64, 64)])
def test_streaming_trace_kernels_give_the_same_image(run, refill):
    """Options stream_run / stream_refill: a wavefront of the shadow and extension kernels owns run x 64 consecutive queue entries and hands
    new rays to its idle lanes whenever `refill` of them are idle.  Which LANE traces a ray is no part of the answer: image, ray counts and
    node visits equal the oracle's and the one-ray-per-lane kernels', with the sorted extension order, frame slots and batches too."""
    w, h = 200, 136
    scene, be, orc = make("soup", w, h, 2500, 6, seed=13, max_path_length=4, frames_in_flight=2, max_batch=3)
    orc.set_option("max_path_length", 4)
    view = scene.view(w, h)
    be.set_option("count_traversal", 1)
    be.set_option("stream_run", 0)              # one ray per lane
    be.render(view)
    plain = be.frame_stats()
    be.reset_accumulation()
    be.set_option("stream_run", run)
    be.set_option("stream_refill", refill)
    for k in range(2):
        be.render(view); orc.render(view)
        if k == 0:
            s = be.frame_stats()
            for key in ("primary_rays", "shadow_rays", "extension_rays", "tris_tested", "instances_entered"):
                assert s[key] == plain[key], key   # every ray does what it did, whichever lane it ran on
            # (a closest-hit lane that waits at a leaf for company meets its later nodes with the SAME ray interval — the order of its own
            # steps does not change — so the node visits are the same too)
            assert s["nodes_visited"] == plain["nodes_visited"]
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.set_option("count_traversal", 0)
    be.set_option("sort_extension_rays", 1)
    views = []
    for i in range(3):
        scene.set_camera([0.3 * i - 0.3, 0.3, -4.0], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    be.render_batch(views)
    for f, v in enumerate(views):
        orc.reset(); orc.render(v)
        assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    with pytest.raises(Exception):
        be.set_option("stream_run", 3)          # runs are powers of two (a 64x64-pixel tile is a whole number of them)
    be.close()


@pytest.mark.parametrize("kind,a,b", [("atrium", 30000, 0), ("soup", 2000, 6), ("gallery", 0, 0)])
def test_packet_trace_option_never_changes_the_image(kind, a, b):
    """Option packet_trace (csrc/traverse_packet.h): camera rays (bit 0, the default) and the camera paths' shadow rays (bit 1) walk the tree as
    wavefront packets on one shared stack, through the eight octant copies of the nodes, instead of one ray per lane.  Which nodes are
    visited, by whom and in which order differs; hits, images and ray counts do not — with rotated instances (the soup: lanes of one packet
    fall into different object-space octants), with and without the traversal counters, for single frames and over frame slots."""
    w, h = 200, 120
    scene, be, orc = make(kind, w, h, a, b, seed=33, max_path_length=3, frames_in_flight=2)
    views = []
    for i in range(2):
        scene.translate_relative([0.05 * i, 0.02 * i, 0.0])
        views.append(scene.view(w, h))
    refs = []
    for v in views:
        orc.reset(); orc.render(v)
        refs.append((orc.accumulator().copy(), orc.stats()["shadow"]))   # (orc.reset() clears the ray counters)
    visits = {}
    for count in (0, 1):
        be.set_option("count_traversal", count)
        for mode in (0, 1, 2, 3, 5):   # (bit 2: packets only for the shadow buckets traced far to near — the directional lights' parallel rays)
            be.set_option("packet_trace", mode)
            for k, v in enumerate(views):   # a new view each: the two frame slots alternate
                be.reset_accumulation()
                be.render(v)
                assert np.array_equal(be.accumulator().view(np.uint32), refs[k][0].view(np.uint32)), (count, mode, k)
                s = be.frame_stats()
                assert s["shadow_rays"] == refs[k][1], (count, mode, k)
                if count:
                    visits[(mode, k)] = tuple(s["nodes_visited"])
    assert visits[(0, 0)] != visits[(1, 0)] and visits[(1, 0)] != visits[(3, 0)]   # they really are different traversals
    be.close()


@pytest.mark.parametrize("kind,a,b", [("atrium", 30000, 0), ("soup", 2000, 4)])
def test_shadow_order_option_never_changes_the_image(kind, a, b):
    """Option shadow_order: which end of a shadow ray the any-hit traversal starts from (default: directional lights far to near, positional
    lights near to far; 1 = every ray near to far, 2 = every ray far to near).  Any hit has no 'right' order: the image, the ray counts
    and the oracle agree under all three; only the node visits differ."""
    w, h = 144, 96
    scene, be, orc = make(kind, w, h, a, b, seed=21, max_path_length=3)
    view = scene.view(w, h)
    orc.render(view)
    ref = orc.accumulator()
    visits = []
    be.set_option("count_traversal", 1)
    for order in (0, 1, 2):
        be.set_option("shadow_order", order)
        be.reset_accumulation()
        be.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), ref.view(np.uint32)), order
        s = be.frame_stats()
        assert s["shadow_rays"] == orc.stats()["shadow"]
        visits.append(s["nodes_visited"][2] if "nodes_visited" in s else None)
    assert visits[0] is None or len(set(visits)) > 1      # the orders are really different traversals
    be.close()


@pytest.mark.parametrize("aperture", [0.04, 0.4])
def test_thin_lens_camera_matches_oracle(aperture):
    """ray_gen.comp:93-135: a lens of nine blades, the lens point drawn per pixel and sample — with the blue-noise tables for the first 256 samples
    and xorshift without them.  Every camera ray then has an origin of its own, which the wavefront packets must not care about
    (option packet_trace 0 / 1 give the same image)."""
    from rfw_rs_amd import HipBackend
    w, h = 136, 72
    scene, be, orc = make("soup", w, h, 2200, 5, seed=41, max_path_length=3, frames_in_flight=2)
    scene.set_camera([0.3, 0.4, -5.0], [0.05, -0.02, 1.0], fov=50.0, aperture=aperture, aspect=w / h)
    view = scene.view(w, h)
    assert view.lens_size > 0.0
    for tables in (False, True):
        if tables:
            t = np.random.default_rng(5).integers(0, 256, 5 * 65536).astype(np.uint32)
            be.set_blue_noise(t); orc.set_blue_noise(t)
        for pk in (1, 0):
            be.set_option("packet_trace", pk)
            be.reset_accumulation(); orc.reset()
            for _ in range(3):
                be.render(view); orc.render(view)
            assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), (aperture, tables, pk)
    be.close()


def test_camera_rays_leave_the_packets_when_the_scene_outgrows_the_caches():
    """While nobody sets option packet_trace, camera rays walk the tree as packets only up to kPacketAutoMaxTriangles (api_internal.h: far
    outside every cache a packet's one dependent scalar load per step loses to 64 independent vector loads per wavefront — bench.py
    --workload atrium32m: 2310 Mrays/s with packets, 3060 without).  The limit is lowered through the environment here (it is read once per
    process: a child process), and the traversal counters tell which kernel ran."""
    import os, subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from rfw_rs_amd import HipBackend, Scene
w, h = 128, 80
scene = Scene().build("soup", 2500, 3, 0.0, 17); scene.set_aspect(w / h)
view = scene.view(w, h)
be = HipBackend.init(w, h, 1.0, max_path_length=2)
be.set_option("count_traversal", 1)
scene.sync(be)
def run():
    be.reset_accumulation(); be.render(view)
    return be.accumulator().copy(), tuple(be.frame_stats()["nodes_visited"])
img_auto, v_auto = run()
copies_auto = be.scene_stats()["packet_copies"]
be.set_option("packet_trace", 0); img0, v0 = run()
# (a scene built beyond the limit carries no packet form of its node copies — 1 KB per node slot: asking for packets afterwards takes
# effect with the next synchronize(), which rebuilds with them)
be.set_option("packet_trace", 1); be.synchronize(); img1, v1 = run()
assert be.scene_stats()["packet_copies"] == 1
assert np.array_equal(img_auto.view(np.uint32), img0.view(np.uint32)) and np.array_equal(img0.view(np.uint32), img1.view(np.uint32))
print("VISITS", v_auto == v0, v_auto == v1, "COPIES", copies_auto)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for limit, want in (("1000", "VISITS True False COPIES 0"), ("100000000", "VISITS False True COPIES 1")):
        env = dict(os.environ, RFW_PACKET_AUTO_MAX_TRIANGLES=limit)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert want in out.stdout, (limit, out.stdout, out.stderr[-500:])
