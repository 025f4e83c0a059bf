"""Host-side cost of one Backend::render call (planning probe, run on the MI355X box: `python tools/probes/host_cost.py`): 2000 calls over
8 frame slots, timed without and with the GPU work.  Round 2: 33-47 us of submission per call (7 launches + events) - 6-8 % of the 0.55 ms
a headline frame takes, so the one-render()-per-frame mode is not host-bound."""
import time, sys
sys.path.insert(0, '.')
from rfw_rs_amd import HipBackend, Scene
scene = Scene().build("atrium", 262267, 0, 0.0, 1)
for (w, h) in ((64, 36), (1920, 1080)):
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=1, frames_in_flight=8)
    scene.mark_all_changed(); scene.sync(be)
    views = []
    for i in range(16):
        scene.set_camera([0.0, 1.6, -8.0 + 0.005 * i], [0.0, 0.0, 1.0], fov=60.0, aspect=w / h)
        views.append(scene.view(w, h))
    for i in range(64): be.render(views[i % 16])
    be.device_synchronize()
    t = time.perf_counter()
    n = 2000
    for i in range(n): be.render(views[i % 16])
    t_submit = time.perf_counter() - t
    be.device_synchronize()
    t_all = time.perf_counter() - t
    print(f"{w}x{h}: host time per render() call {t_submit / n * 1e6:.1f} us (submission only), {t_all / n * 1e6:.1f} us per frame with the GPU work")
    be.close()
