"""Would cutting the shadow wavefronts at K trips and finishing the unfinished rays in a second, compacted pass pay?  (round 5 planning)

A wavefront of k_shadow issues instructions for as many trips as its LONGEST lane needs (node-test lane utilisation 0.57, finished lanes
0.66 of the longest).  This probe takes the REAL shadow queue of one 1080p frame of the bench scene (debug_read sh_o / sh_d: 8 buckets, the
order k_shadow walks them in), asks the device for the nodes every ray visits (rfw_hip_debug_occludes_depth — near-to-far for every ray:
the far-to-near order of the directional bucket is not modelled) and compares, in node visits issued per wavefront-lane:
    now      sum over wavefronts of 64 consecutive entries of max(nodes)
    cut K    pass 1: sum of min(max(nodes), K); pass 2: the rays with more than K nodes, compacted in queue order, RESUMED: sum of max(nodes - K)
             (restart instead of resume: sum of max(nodes))
usage (GPU box): python3 tools/probes/tail_model.py [triangles]"""
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
be = HipBackend.init(w, h, 1.0, max_path_length=1)
scene.sync(be)
be.render(scene.view(w, h))
cap = w * h  # (capacity of one bucket's region = the frame's paths; world = 1)
raw = be.debug_read("counters", 4096)
counts = np.frombuffer(raw.tobytes()[32:32 + 8 * 8 * 4], dtype=np.uint32).reshape(8, 8)[0]  # shadow[bounce 0][bucket]
cap = len(be.debug_read("sh_o", 1 << 34)) // 16 // 8
so = be.debug_read("sh_o", cap * 16 * 8).view(np.float32).reshape(8, cap, 4)
sd = be.debug_read("sh_d", cap * 16 * 8).view(np.float32).reshape(8, cap, 4)
out = {"rays": int(counts.sum()), "per_bucket": [int(c) for c in counts]}
depths = []
for b in range(7, -1, -1):  # the order k_shadow walks the buckets in
    n = int(counts[b])
    if not n:
        continue
    o, d, tm = so[b, :n, :3].copy(), sd[b, :n, :3].copy(), np.minimum(sd[b, :n, 3] - np.float32(1e-4), np.float32(3e38)).astype(np.float32)
    occ, dep = be.occludes_depth(o, d, tm)
    depths.append((b, dep.astype(np.int64), occ))
    out.setdefault("occluded_share", {})[b] = round(float(occ.mean()), 3)


def waves_max(x):
    pad = (-len(x)) % 64
    return np.pad(x, (0, pad)).reshape(-1, 64).max(axis=1)


now = sum(int(waves_max(dep).sum()) for _, dep, _ in depths)
mean = sum(int(dep.sum()) for _, dep, _ in depths) / out["rays"]
out["nodes_per_ray"] = round(mean, 2)
out["now_wave_trips_per_ray"] = round(now / (out["rays"] / 64) / 64, 2)  # = mean of the per-wavefront maxima / 1 ... per lane
out["now_lane_trips"] = now * 64
res = {}
for K in (8, 12, 16, 20, 24, 28, 32, 40):
    p1 = sum(int(np.minimum(waves_max(dep), K).sum()) for _, dep, _ in depths)
    long_resume = np.concatenate([dep[dep > K] - K for _, dep, _ in depths])
    long_restart = np.concatenate([dep[dep > K] for _, dep, _ in depths])
    p2r = int(waves_max(long_resume).sum()) if len(long_resume) else 0
    p2s = int(waves_max(long_restart).sum()) if len(long_restart) else 0
    res[K] = {"unfinished_share": round(len(long_resume) / out["rays"], 3), "resume_vs_now": round((p1 + p2r) / now, 3), "restart_vs_now": round((p1 + p2s) / now, 3),
              "pass1_vs_now": round(p1 / now, 3)}
out["cut"] = res
print(json.dumps(out))
be.close()
