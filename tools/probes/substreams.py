"""One frame at a time on a single instance: does splitting the frame's tiles over sub-streams pay? (planning probe)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rfw_rs_amd import HipBackend, Scene
w, h = 1920, 1080
scene = Scene().build("atrium", 1048576, 0, 0.0, 0xC0FFEE)
scene.set_aspect(w / h)
views = []
for i in range(16):
    scene.set_camera([0.2 * i - 1.0, 1.6, -6.0], [0.0, 0.0, 1.0], fov=60.0, aspect=w / h)
    views.append(scene.view(w, h))
for streams in (0, 2, 3, 4):
    be = HipBackend.init(w, h, 1.0, streams=streams)
    scene.mark_all_changed(); scene.sync(be)
    for i in range(20):
        be.render(views[i % 16]); be.device_synchronize()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        be.render(views[i % 16]); be.device_synchronize()
    dt = (time.perf_counter() - t0) / n
    st = be.frame_stats()
    print(f"streams {streams}: {dt*1e3:.4f} ms per frame")
    be.close()
