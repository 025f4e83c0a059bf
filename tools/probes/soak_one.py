"""One configuration of tests/soak_gpu.py, check by check (debugging aid): python3 tools/probes/soak_one.py <builder> <frames_in_flight> [tris inst seed w h]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle.bindings import Oracle
from rfw_rs_amd import HipBackend, Scene
builder, fif = int(sys.argv[1]), int(sys.argv[2])
tris, inst, seed, w, h = (int(x) for x in (sys.argv[3:8] if len(sys.argv) >= 8 else (1732, 10, 99, 96, 70)))
rng = np.random.default_rng(1)
scene = Scene().build("soup", tris, inst, 0.0, seed); scene.set_aspect(w / h)
be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder, frames_in_flight=fif)
orc = Oracle(w, h, threads=8, max_path_length=3)
for key, val in (("packet_trace", os.environ.get("PK")),):
    if val is not None:
        be.set_option(key, int(val))
if os.environ.get("SORT"):
    be.set_option("sort_extension_rays", int(os.environ["SORT"]))
scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
if os.environ.get("RECOLOUR"):
    scene.recolour_material(2, [200, 40, 90], 120)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
o = rng.uniform(-5, 5, (20000, 3)).astype(np.float32); d = rng.normal(size=(20000, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
g, r = be.intersect(o, d), orc.intersect(o, d)
print("intersect inst", np.array_equal(g["inst"], r["inst"]), "tri", np.array_equal(g["tri"], r["tri"]), "t", np.array_equal(g["t"][r["inst"] >= 0].view(np.uint32), r["t"][r["inst"] >= 0].view(np.uint32)),
      "differing rays", int((g["inst"] != r["inst"]).sum()))
tm = rng.uniform(0.05, 9.0, 20000).astype(np.float32)
print("occludes", np.array_equal(be.occludes(o, d, tm), orc.occludes(o, d, tm)))
if os.environ.get("APERTURE"):
    scene.set_camera([0.3, 0.4, -5.0], [0.05, -0.02, 1.0], fov=50.0, aperture=float(os.environ["APERTURE"]), aspect=w / h)
view = scene.view(w, h)
print("lens_size", view.lens_size)
if os.environ.get("BACK_TO_BACK"):
    for _ in range(2):
        be.render(view); orc.render(view)
    a, b = be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)
    print("two frames back to back: accumulator equal", np.array_equal(a, b), "differing pixels", int((a != b).any(axis=-1).sum()))
for k in range(3):
    be.render(view); orc.render(view)
    a, b = be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)
    print("frame", k, "accumulator equal", np.array_equal(a, b), "differing pixels", int((a != b).any(axis=-1).sum()))
be.close()
