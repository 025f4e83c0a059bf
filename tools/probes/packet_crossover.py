"""Where do wavefront packets stop paying?  Frames per second of the headline configuration (1080p, primary + shadow, 12 frame slots) with
camera rays as packets (option packet_trace = 1) and one per lane (0), over scene sizes from in-cache to far outside every cache."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rfw_rs_amd import HipBackend, Scene
w, h = 1920, 1080
sizes = [int(x) for x in sys.argv[1:]] or [1048576, 2097152, 4194304, 8388608, 16777216]
for tris in sizes:
    scene = Scene().build("atrium", tris, 0, 0.0, 0xC0FFEE)
    scene.set_aspect(w / h)
    from bench import dolly_views                   # the bench's own views: the scene's camera, 5 mm dolly per frame
    views = dolly_views(scene.view(w, h), 16, 0.005)
    be = HipBackend.init(w, h, 1.0, frames_in_flight=12)
    scene.sync(be)
    st = be.scene_stats()
    out = {}
    for pk in (1, 0, 1, 0):
        be.set_option("packet_trace", pk)
        for i in range(24):
            be.render(views[i % 16])
        be.device_synchronize()
        n = 120
        t0 = time.perf_counter()
        for i in range(n):
            be.render(views[i % 16])
        be.device_synchronize()
        out.setdefault(pk, []).append(round((time.perf_counter() - t0) / n * 1e3, 4))
    print(f"{st['triangles']} triangles, {st['blas_nodes']} nodes: ms per frame packets {out[1]}  one ray per lane {out[0]}", flush=True)
    be.close()
