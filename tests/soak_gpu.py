"""Randomised differential soak, GPU against the oracle (not collected by pytest; run on an MI355X: `ITERS=300 python tests/soak_gpu.py`).
Random soups x instance counts x builders x frame slots x frame batches x odd resolutions: ray queries (closest / any hit, incl. axis-parallel rays) and two
accumulated frames must be bit-identical; then two large atrium scenes.  Round 1: 300 + 2 configurations, then 200 + 2 and, on the final kernels of the round, 300 + 2 more with frame batches and downloads in the mix: 0 mismatches.
Round 2 (blue-noise tables, sorted extension rays, single-material edits with `changed` bits and rfw_hip_render_samples in the mix): 250 + 2 configurations, 0 mismatches;
on the final build of the round 300 + 2 more and 60 random times of an animated, twice-instantiated glTF document: 0 mismatches.
Round 4 (packets, per-octant node copies, heads-first uploads): the first 200 configurations found 26 mismatching images, every one of them host
builder x three frame slots — the per-slot TLAS of the host builder was uploaded on the owner's stream and expanded into its octant copies on the
slot's, visible only in a process whose earlier backends had dirtied the recycled allocation (now a GPU test:
test_backends_one_after_the_other_in_one_process).  After the fix 300 + 2 + 60: 0 mismatches.  A failing configuration now names its failed checks.
Round 5 (spatial splits, the triangle test without early outs): every fourth configuration an atrium with split walls under a random threshold;
at the end of the round (flat traversal loops, streaming ones too; k_shade in workgroups of 256 or 512) streaming runs / refills and the shade
group are random too.
Round 6 (queue counters in a ring, the TLAS in one workgroup, 16-byte aligned node types): the TLAS path is random too."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.bindings import Oracle
from rfw_rs_amd import HipBackend, Scene
rng = np.random.default_rng(12345)
bad = 0
t0 = time.time()
for it in range(int(os.environ.get("ITERS", "24"))):
    tris = int(rng.integers(200, 6000)); inst = int(rng.integers(1, 24)); seed = int(rng.integers(1, 1 << 30))
    builder = int(rng.integers(0, 4)); fif = int(rng.choice([0, 3])); w, h = int(rng.choice([64, 96, 130])), int(rng.choice([48, 70]))
    # round 5: every fourth configuration is an atrium of 9 000 ... 50 000 triangles (walls and floors of two triangles: spatial splits, with
    # duplicates whose packets must report their originals), under a random split threshold
    big = int(rng.integers(0, 4)) == 0
    if big:
        tris = int(rng.integers(9000, 50000))
        scene = Scene().build("atrium", tris, int(rng.integers(0, 2)), 0.0, seed)
    else:
        scene = Scene().build("soup", tris, inst, 0.0, seed)
    scene.set_aspect(w / h)
    mb = int(rng.choice([0, 5]))
    be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder, frames_in_flight=fif, max_batch=mb)
    be.set_option("spatial_splits", float(rng.choice([0.0, 8e-5, 1e-6])))
    be.set_option("tlas_fused", int(rng.choice([0, 1, 2])))   # round 6: the one-workgroup TLAS build (2 ... 16 384 instances) or the launch chain
    orc = Oracle(w, h, threads=8, max_path_length=3)
    # round 2: the blue-noise sampler with seeded tables, extension rays in sorted order, a material edit with a `changed` bit
    bn = int(rng.integers(0, 3))
    if bn:
        t = np.random.default_rng(seed).integers(0, 256, 5 * 65536).astype(np.uint32)
        be.set_blue_noise(t); orc.set_blue_noise(t)
    be.set_option("sort_extension_rays", int(rng.integers(0, 3)))
    # round 5 (flat loops, k_shade's workgroup size by situation): the streaming kernels forced on or off for any number of frame slots, with
    # short runs and early refills in the mix; the shading kernel's workgroup size left to the rule or forced
    sr = int(rng.choice([-1, 0, 1, 2, 8]))
    if sr >= 0:
        be.set_option("stream_run", sr)
        be.set_option("stream_refill", int(rng.choice([4, 12, 40])))
    be.set_option("shade_group", int(rng.choice([0, 256, 512])))
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    if rng.integers(0, 2):
        scene.recolour_material(int(rng.integers(0, 6)), [int(x) for x in rng.integers(20, 250, 3)], int(rng.integers(10, 250)))
        scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o = rng.uniform(-5, 5, (20000, 3)).astype(np.float32) * (2.5 if big else 1.0); d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 50)] * rng.choice([-1, 1], (50, 1))  # axis-parallel
    g, r = be.intersect(o, d), orc.intersect(o, d)
    failed = []   # names of the checks that failed
    ok = np.array_equal(g["inst"], r["inst"]) and np.array_equal(g["tri"], r["tri"]) and np.array_equal(g["t"][r["inst"] >= 0].view(np.uint32), r["t"][r["inst"] >= 0].view(np.uint32))
    if not ok: failed.append("intersect")
    tm = rng.uniform(0.05, 9.0, 20000).astype(np.float32)
    c = np.array_equal(be.occludes(o, d, tm), orc.occludes(o, d, tm)); ok = ok and c
    if not c: failed.append("occludes")
    if rng.integers(0, 3) == 0:   # round 4: a thin-lens camera (every camera ray of a packet has its own origin)
        scene.set_camera([0.3, 0.4, -5.0], [0.05, -0.02, 1.0], fov=50.0, aperture=float(rng.choice([0.03, 0.3])), aspect=w / h)
    view = scene.view(w, h)
    for _ in range(2):
        be.render(view); orc.render(view)
    c = np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)); ok = ok and c
    if not c: failed.append("two frames")
    if mb:  # a batch of independent views in one launch per stage: every frame == the oracle's render of that view
        views = []
        for k in range(int(rng.integers(2, mb + 1))):
            scene.set_camera(list(rng.uniform(-1.5, 1.5, 3) + np.array([0, 0, -4.0])), [float(rng.uniform(-0.2, 0.2)), 0.0, 1.0], fov=45.0, aspect=w / h)
            views.append(scene.view(w, h))
        be.render_batch(views)
        host = be.host_frame()
        for k, v in enumerate(views):
            orc.reset(); orc.render(v)
            c = np.array_equal(be.accumulator_at(k).view(np.uint32), orc.accumulator().view(np.uint32)); ok = ok and c
            if not c: failed.append(f"batch frame {k}")
            be.download_frame(host, frame=k); be.wait_downloads(host)
            c = np.array_equal(host.view(np.uint32), orc.framebuffer().view(np.uint32)); ok = ok and c
            if not c: failed.append(f"batch download {k}")
    if mb:  # k samples of one image in one launch per stage: the sum of the per-sample images, i.e. k render() calls up to rounding
        k = int(rng.integers(2, mb + 1))
        be.render_samples(view, k)     # a different call sequence after the batch above: a new image of `view`
        orc.reset()
        for _ in range(k):
            orc.render(view)
        ga, ra = be.accumulator().astype(np.float64), orc.accumulator().astype(np.float64)
        c = np.linalg.norm(ga - ra) <= 1e-6 * max(np.linalg.norm(ra), 1e-30); ok = ok and c
        if not c: failed.append("samples")
    s, so = be.frame_stats(), orc.stats()
    print(it, "tris", tris, "batch", mb, "inst", inst, "builder", builder, "fif", fif, f"{w}x{h}", "bn", bn, "OK" if ok else "MISMATCH " + ", ".join(failed), flush=True)
    bad += 0 if ok else 1
    be.close()
print("mismatches:", bad, "time", round(time.time() - t0, 1))
# a glTF document's own animation at random times (skinning + refit + TLAS on the device every step), two graphs of it
import pathlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gltf_util import write_animated_gltf
tmp = pathlib.Path(tempfile.mkdtemp())
scene = Scene().load_gltf(str(write_animated_gltf(tmp)))
g = scene.instantiate_graph()
scene.set_graph_transform(g, translation=(1.2, 0.0, -0.8), rotation=(0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))), scale=(0.8, 0.8, 0.8))
w, h = 112, 80
scene.set_aspect(w / h)
view = scene.view(w, h)
be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=3)
orc = Oracle(w, h, threads=8, max_path_length=3)
abad = 0
for it in range(int(os.environ.get("ANIM_ITERS", "12"))):
    scene.set_animation_time(float(rng.uniform(-1.0, 7.0)))
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    be.reset_accumulation(); orc.reset()
    be.render(view); orc.render(view)
    ok = np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    abad += 0 if ok else 1
print("animated glTF mismatches:", abad, flush=True)
be.close()
# large scenes: hits and one frame
for it, tris in enumerate((60000, 262267)):
    scene = Scene().build("atrium", tris, 0, 0.0, 77 + it); w, h = 320, 180; scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=2)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o = rng.uniform(-8, 8, (50000, 3)).astype(np.float32); o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.2
    d = rng.normal(size=(50000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    g, r = be.intersect(o, d), orc.intersect(o, d)
    ok = np.array_equal(g["inst"], r["inst"]) and np.array_equal(g["tri"], r["tri"]) and np.array_equal(g["t"][r["inst"] >= 0].view(np.uint32), r["t"][r["inst"] >= 0].view(np.uint32))
    view = scene.view(w, h)
    be.render(view); orc.render(view)
    ok = ok and np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    print("large", tris, "OK" if ok else "MISMATCH", (g["inst"] >= 0).mean(), flush=True)
    be.close()
# round 4: re-sent meshes (registered host copies, heads-first uploads on two streams, incremental edits from registered copies) with frames in
# flight on frame slots — every image against a single-slot backend that is given the same scene state synchronously
scene = Scene().build("atrium", 700000, 1, 0.0, 21)
w, h = 160, 90
scene.set_aspect(w / h)
views = []
for i in range(4):
    scene.set_camera([0.6 * i - 1.0, 1.6, -6.0 + 0.3 * i], [0.05 * i, 0.0, 1.0], fov=55.0, aspect=w / h)
    views.append(scene.view(w, h))
be = HipBackend.init(w, h, 1.0, max_path_length=2, frames_in_flight=3)
ref = HipBackend.init(w, h, 1.0, max_path_length=2)
scene.sync(be); scene.mark_all_changed(); scene.sync(ref)
rbad = 0
def edit(kind):
    if kind == 0:
        scene.mark_all_changed()                                   # everything re-sent: the full build, heads first
    elif kind == 1:
        scene.replace_mesh_with_sphere(int(rng.integers(1, 65)), int(rng.integers(2, 5)), int(rng.integers(1, 1000)))   # one mesh: incremental
    elif kind == 2:
        for _ in range(int(rng.integers(2, 6))):
            scene.replace_mesh_with_sphere(int(rng.integers(1, 65)), int(rng.integers(2, 5)), int(rng.integers(1, 1000)))
    else:
        scene.replace_mesh_with_sphere(int(rng.integers(1, 65)), int(rng.integers(2, 5)), int(rng.integers(1, 1000)))
        scene.mark_all_changed()
pending = None   # (iteration, kind, host frames of `be`, reference frames): compared one iteration LATER, after the next synchronize has been issued
n_resend = int(os.environ.get("RESEND_ITERS", "40"))
for it in range(n_resend + 1):
    if it < n_resend:
        kind = int(rng.integers(0, 4))
        edit(kind)
        scene.sync(be)                                             # the previous iteration's frames may still be in flight on the slots
        frames = [be.host_frame() for _ in views]
        for v, dst in zip(views, frames):
            be.render(v); be.download_frame(dst)                   # nothing waits
    if pending is not None:
        pit, pkind, pframes, prefs = pending
        be.wait_downloads()
        okr = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(pframes, prefs))
        for dst in pframes:
            be.free_host_frame(dst)
        print("resend", pit, "kind", pkind, "OK" if okr else "MISMATCH", flush=True)
        rbad += 0 if okr else 1
        pending = None
    if it < n_resend:
        scene.mark_all_changed(); scene.sync(ref)
        refs = []
        for v in views:
            ref.reset_accumulation(); ref.render(v); ref.device_synchronize()
            refs.append(ref.framebuffer().copy())
        pending = (it, kind, frames, refs)
print("re-sent meshes with frames in flight, mismatches:", rbad)
be.close(); ref.close()
