"""Occluder-broadcast experiment (traverse.h, RFW_SHADOW_BROADCAST): counters of one instrumented frame of the bench scene.
usage (GPU box): RFW_HIP_LIB=.../librfw_hip_bc.so python3 tools/probes/bc_probe.py [triangles]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rfw_rs_amd import HipBackend, Scene  # noqa: E402

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
w, h = 1920, 1080
scene = Scene().build("atrium", tris, 0, 0.0, 0xC0FFEE)
scene.set_aspect(w / h)
be = HipBackend.init(w, h, 1.0, max_path_length=int(os.environ.get("MPL", "1")))
scene.sync(be)
be.set_option("count_traversal", 1)
be.render(scene.view(w, h))
s = be.frame_stats()
raw = be.debug_read("counters", 4096)
u64 = np.frombuffer(raw.tobytes()[: len(raw) // 8 * 8], dtype=np.uint64)
# QueueCounters: ext[8] u32, shadow[8][8] u32 = 288 B; trav[3][3] u64 (72 B) @288; overflow, pad @360; wave_max_nodes[3] @376; wave_exec[3][2] @400; max_nodes[3] @448; pad2 @472; wave_uniform[3] @480; pad3 @504
pad2, pad3 = int(u64[472 // 8]), int(u64[504 // 8])
out = {"shadow_rays": s["shadow_rays"], "nodes_per_shadow_ray": s["nodes_visited"][2] / max(s["shadow_rays"], 1), "tris_per_shadow_ray": s["tris_tested"][2] / max(s["shadow_rays"], 1),
       "node_test_executions_per_wave": s["node_test_executions"][2] / max(s["shadow_rays"] / 64, 1), "tri_test_executions_per_wave": s["tri_test_executions"][2] / max(s["shadow_rays"] / 64, 1),
       "broadcasts": pad2, "lanes_retired_by_broadcast": pad3, "broadcasts_per_wave": pad2 / max(s["shadow_rays"] / 64, 1), "retired_per_broadcast": pad3 / max(pad2, 1),
       "retired_share_of_shadow_rays": pad3 / max(s["shadow_rays"], 1)}
print(json.dumps(out))
be.close()
