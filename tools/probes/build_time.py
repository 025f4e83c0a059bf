"""Device BVH builder timing (planning / DESIGN numbers, not a test): full rebuild of the bench scene's meshes, warm, and one small mesh
of C4-as-65-meshes replaced.  usage (GPU box): python3 tools/probes/build_time.py [builder]"""
import json
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rfw_rs_amd import HipBackend, Scene  # noqa: E402

builder = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only = sys.argv[2] if len(sys.argv) > 2 else None
out = {"builder": builder}
for name, tris, sep in (("atrium1m_2_meshes", 1048576, 0), ("atrium1m_65_meshes", 1048576, 1), ("atrium262k", 262267, 0)):
    if only and only != name:
        continue
    scene = Scene().build("atrium", tris, sep, 0.0, 0xC0FFEE)
    be = HipBackend.init(64, 64, 1.0, builder=builder)
    t0 = time.perf_counter(); scene.sync(be); be.device_synchronize(); cold = time.perf_counter() - t0
    warm = []
    for _ in range(4):
        scene.mark_all_changed()
        t0 = time.perf_counter(); scene.sync(be); be.device_synchronize(); warm.append(time.perf_counter() - t0)
    st = be.scene_stats()
    out[name] = {"triangles": st["triangles"], "blas_nodes": st["blas_nodes"], "cold_ms": round(cold * 1e3, 2), "warm_sync_ms": [round(x * 1e3, 2) for x in warm],
                 "reported_blas_build_ms": round(st["ms_blas_build"], 2)}
    if sep:
        one = []
        for k in range(4):
            scene.replace_mesh_with_sphere(3 + k, 2 + k, 77 + k)   # one 5120-triangle sphere mesh replaced (same size, other displacement)
            t0 = time.perf_counter(); scene.sync(be); host = time.perf_counter() - t0; be.device_synchronize(); one.append((round(host * 1e3, 3), round((time.perf_counter() - t0) * 1e3, 3)))
        out[name]["one_mesh_edit_ms_host_and_device_done"] = one
    be.close()
print(json.dumps(out))
