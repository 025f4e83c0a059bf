import sys, time
sys.path.insert(0, "/root/repo")
from rfw_rs_amd import HipBackend, Scene
w, h = 1920, 1080
scene = Scene().build("atrium", 1048576, 0, 0.0, 0xC0FFEE)
scene.set_aspect(w / h)
views = []
for i in range(16):
    scene.set_camera([0.2 * i - 1.0, 1.6, -6.0], [0.0, 0.0, 1.0], fov=60.0, aspect=w / h)
    views.append(scene.view(w, h))
be = HipBackend.init(w, h, 1.0, frames_in_flight=12)
scene.sync(be)
for i in range(24):
    be.render(views[i % 16])
be.device_synchronize()
for n in (20, 200):
    t0 = time.perf_counter()
    for i in range(n):
        be.render(views[i % 16])
    t1 = time.perf_counter()
    be.device_synchronize()
    t2 = time.perf_counter()
    print(f"{n} frames: host enqueue {1e3*(t1-t0)/n:.4f} ms/frame, total {1e3*(t2-t0)/n:.4f} ms/frame")
be.close()
