"""Two backends one after the other in ONE process (recycled device memory): python3 tools/probes/soak_two.py b1 f1 b2 f2"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle.bindings import Oracle
from rfw_rs_amd import HipBackend, Scene
args = [int(x) for x in sys.argv[1:]]
cfgs = list(zip(args[0::2], args[1::2]))
rng = np.random.default_rng(1)
for n, (builder, fif) in enumerate(cfgs):
    tris, inst, seed, w, h = int(rng.integers(200, 6000)), int(rng.integers(1, 24)), int(rng.integers(1, 1 << 30)), 96, 70
    scene = Scene().build("soup", tris, inst, 0.0, seed); scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder, frames_in_flight=fif)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    view = scene.view(w, h)
    for _ in range(2):
        be.render(view); orc.render(view)
    a, b = be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)
    print(n, "builder", builder, "fif", fif, "tris", tris, "inst", inst, "two frames equal", np.array_equal(a, b), "differing pixels", int((a != b).any(axis=-1).sum()), flush=True)
    be.close()
