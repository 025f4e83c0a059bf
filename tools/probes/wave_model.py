"""Wave-occupancy model from REAL per-ray traversal lengths (planning probe, run on the MI355X box: `python tools/probes/wave_model.py`).

The traversal kernels lose lanes as rays finish at different times (`finished_lanes_bound` in the bench line).  This probe measures how
much a scheme that re-packs unfinished rays could win at best, before anyone writes it: it traces the bench scene's primary rays
(closest hit, rfw_hip_depth_test) and the shadow rays of a rendered frame, read back from its shadow queue (any hit,
rfw_hip_debug_occludes_depth), takes the 4-wide nodes each ray visited as its length, forms wavefronts the way the kernels do
(8 x 8 pixel blocks; shadow rays per light in block order) and counts wave-iterations

  * as the kernels run today: a wavefront runs until its longest ray is done (sum of the per-wave maxima),
  * with an early exit: when fewer than T lanes are still active the wavefront stops and its unfinished rays continue in new, densely
    packed wavefronts (again with early exit, recursively) - state save / restore costed at C iterations per continued ray-wave; or
    ("restart") the unfinished rays are simply traced again from their start in those new wavefronts, no state saved,
  * with ideal refill: every lane always busy (sum of lengths / 64): the bound no scheme can beat.

Prints one JSON object.  Nothing here is part of the product or of the tests; the oracle is not used."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rfw_rs_amd import HipBackend, Scene  # noqa: E402


def waves_of_blocks(values, w, h):
    """(h, w) per-pixel values -> (n_waves, 64): one row per 8 x 8 pixel block (ragged edges dropped)."""
    hh, ww = h // 8 * 8, w // 8 * 8
    v = values[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    return v


def early_exit_cost(lengths, threshold, overhead, restart=False):
    """lengths: (n_waves, 64) with 0 for empty lanes.  Returns wave-iterations when a wavefront exits as soon as fewer than `threshold`
    lanes are active and the leftovers are re-packed (sorted by remaining length is NOT assumed: queue order)."""
    total = 0.0
    cur = lengths.astype(np.int64)
    for _ in range(12):
        if cur.size == 0:
            break
        s = -np.sort(-cur, axis=1)                      # per wave, longest first
        # the wave runs until the (threshold)-th longest ray finishes: after that fewer than `threshold` lanes are active
        run = s[:, threshold - 1] if threshold <= 64 else s[:, 0]
        run = np.where(s[:, 0] > 0, np.maximum(run, 0), 0)
        full = s[:, 0]
        stop = np.where(run > 0, run, full)             # waves with fewer than `threshold` rays in total run to the end
        total += float(stop.sum())
        left = np.maximum(cur - stop[:, None], 0)
        left = cur[left > 0] if restart else left[left > 0]   # restart: an unfinished ray is traced again from its start (no state is saved)
        if left.size == 0:
            break
        total += overhead * (left.size / 64.0)          # state save + restore, per continued wavefront's worth of rays
        pad = (-left.size) % 64
        cur = np.concatenate([left, np.zeros(pad, np.int64)]).reshape(-1, 64)
        if cur.shape[0] <= 2:                           # the last few wavefronts simply run out
            total += float(cur.max(axis=1).sum())
            break
    return total


def main():
    w, h = 1920, 1080
    tris = int(os.environ.get("TRIS", "1048576"))
    scene = Scene().build("atrium", tris, 0, 0.0, 0xC0FFEE)
    scene.set_aspect(w / h)
    v = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=1)   # a full-size instance: the query calls work in chunks of its ray capacity
    scene.mark_all_changed(); scene.sync(be)
    # pinhole rays through the pixel centres (generate_eye_ray without jitter and lens)
    px, py = np.meshgrid(np.arange(w, dtype=np.float32) + 0.5, np.arange(h, dtype=np.float32) + 0.5)
    pos = np.array([v.pos.x, v.pos.y, v.pos.z], np.float32)
    p1 = np.array([v.p1.x, v.p1.y, v.p1.z], np.float32)
    right = np.array([v.right.x, v.right.y, v.right.z], np.float32)
    up = np.array([v.up.x, v.up.y, v.up.z], np.float32)
    target = p1 + (px / w)[..., None] * right + (py / h)[..., None] * up
    d = target - pos
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    o = np.broadcast_to(pos, d.shape)
    print("primary rays", file=sys.stderr, flush=True)
    hits, depth = be.depth_test(o.reshape(-1, 3), d.reshape(-1, 3))
    depth = depth.reshape(h, w).astype(np.int64)
    out = {"scene_triangles": scene.triangle_count, "resolution": [w, h]}

    def report(name, waves):
        waves = waves[waves.max(axis=1) > 0]
        today = float(waves.max(axis=1).sum())
        ideal = float(waves.sum()) / 64.0
        r = {"wavefronts": int(waves.shape[0]), "mean_length": round(float(waves[waves > 0].mean()), 2), "wave_iterations_today": today,
             "finished_lanes_bound": round(ideal / today, 3), "ideal_refill_gain": round(1.0 - ideal / today, 3)}
        for t in (8, 16, 24, 32):
            for c in (2.0, 6.0):
                r[f"early_exit_T{t}_overhead{int(c)}"] = round(1.0 - early_exit_cost(waves, t, c) / today, 3)
            r[f"early_exit_restart_T{t}"] = round(1.0 - early_exit_cost(waves, t, 0.5, restart=True) / today, 3)
        out[name] = r

    report("primary_closest_hit", waves_of_blocks(depth, w, h))
    # shadow rays: the REAL ones of a rendered frame (next-event estimation towards a light picked at random per pixel), read back from the
    # frame's shadow queue: one region per light bucket, 64 consecutive entries = one wavefront of k_shadow
    be.render(v)
    cap = (w + 63) // 64 * ((h + 63) // 64) * 4096
    ctr = np.frombuffer(be.debug_read("counters", 8 * 4 + 8 * 8 * 4).tobytes(), np.uint32)
    counts = ctr[8:16]                                   # shadow[bounce 0][bucket]
    sh_o = np.frombuffer(be.debug_read("sh_o", cap * 16 * 8).tobytes(), np.float32).reshape(-1, 4)
    sh_d = np.frombuffer(be.debug_read("sh_d", cap * 16 * 8).tobytes(), np.float32).reshape(-1, 4)
    for bucket, cnt in enumerate(counts):
        cnt = int(cnt)
        if cnt == 0:
            continue
        print("shadow rays of bucket", bucket, cnt, file=sys.stderr, flush=True)
        ro, rd = sh_o[bucket * cap: bucket * cap + cnt], sh_d[bucket * cap: bucket * cap + cnt]
        occ, dep = be.occludes_depth(ro[:, :3].copy(), rd[:, :3].copy(), rd[:, 3] - np.float32(1e-4))
        dep = dep.astype(np.int64)
        cnt = int(dep.size)
        pad = (-cnt) % 64
        waves = np.concatenate([dep, np.zeros(pad, np.int64)]).reshape(-1, 64)
        name = f"shadow_any_hit_bucket{bucket}"
        report(name, waves)
        out[name]["rays"] = cnt
        out[name]["occluded_fraction"] = round(float(occ.mean()), 3)
        out[name]["mean_length_occluded"] = round(float(dep[occ].mean()), 2) if occ.any() else None
        out[name]["mean_length_unoccluded"] = round(float(dep[~occ].mean()), 2) if (~occ).any() else None
    be.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
