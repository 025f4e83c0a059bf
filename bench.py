#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + shadow, 1 spp) of the hot path on N MI355X (contract in the task statement).

A step = one render() of one 1920x1080 frame of the ~1M-triangle synthetic atrium scene (the configuration the
metric's target is quoted on, BASELINE.json north_star; SURVEY.md §8d C4 geometry, primary + shadow rays), all
inputs resident in HBM.  For N > 1 the frame is sharded by 64x64 tiles across ranks and the accumulator slabs are
all-gathered with RCCL once per frame (strong scaling: the frame is fixed)."""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Every renderer instance launches on its own HIP stream, and the runtime deals streams to hardware queues (4 by default): two
# instances whose streams land on one queue serialise.  Ask for enough queues BEFORE the HIP runtime starts (it reads this once).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if int(os.environ.get("WORLD_SIZE", "1")) > 2 else "16")  # N > 2: 12 instance streams + RCCL's

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="atrium1m", choices=["atrium1m", "atrium262k", "cornell", "spheres10k"],
                    help="atrium1m = headline (C4 geometry, primary+shadow); atrium262k = C2; spheres10k = C3 (atrium262k + 10 000 "
                         "animated icosphere instances, synchronize() every frame); cornell = C1 geometry")
    ap.add_argument("--max-path-length", type=int, default=1, help="1 = primary+shadow (the metric); 3 = the reference's path tracer (C4)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="renderer instances used round-robin, each on its own stream: frame k+1 is traced while frame k's tail, "
                         "all-gather and assemble finish (1 = strictly one frame at a time)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames traced per rfw_hip_render_batch call (one launch per stage and, with a sharded frame, ONE all-gather for the "
                         "whole batch); 1 = one render() per frame")
    ap.add_argument("--procedural", action="store_true",
                    help="hand the generated scene to the backend directly; default: write it as a binary glTF 2.0 file (host/gltf_export.cpp) and "
                         "run on what the glTF importer (host/gltf.cpp) reads back — the configurations of BASELINE.json are glTF scenes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="single-GPU study of multi-GPU scaling: render only rank 0's tiles of an N-way tile shard (and de-tile a stand-in "
                         "gathered buffer), without any collective; the value then counts this shard's rays only")
    ap.add_argument("--readback", nargs="?", const="float", default=None, choices=["float", "presented"],
                    help="copy every finished frame to (pinned) host memory inside the timed region, queued behind its kernels (the PCIe-inclusive "
                         "rate of DESIGN.md; never the headline value): the RGBA32F frame, or the presented BGRA8 sRGB frame of the reference's swap chain")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    # RFW_BENCH_DIST_BACKEND=gloo is a TEST HOOK: it lets two ranks share one GPU (RCCL refuses duplicate devices) so the
    # N > 1 code path can be exercised on a 1-GPU box; slabs are then staged through host memory.  Default: nccl (= RCCL).
    dist_backend = os.environ.get("RFW_BENCH_DIST_BACKEND", "nccl")
    dev = (local_rank % torch.cuda.device_count()) if world > 1 else 0
    torch.cuda.set_device(dev)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(dist_backend)

    from rfw_rs_amd import HipBackend, Scene

    w, h = args.width, args.height
    tris = {"atrium1m": 1048576, "atrium262k": 262267, "cornell": 0, "spheres10k": 262267}[args.workload]
    scene = Scene().build("cornell") if args.workload == "cornell" else Scene().build("atrium", tris, 0, 0.0, 0xC0FFEE)
    scene_source = "procedural"
    if not args.procedural:
        # the synthetic scene as a real glTF file, read back through the importer (a lossless round trip: tests/test_gltf.py)
        import tempfile
        with tempfile.TemporaryDirectory(prefix="rfw_bench_") as tmp:
            t_io = time.time()
            glb = scene.save_glb(os.path.join(tmp, f"{args.workload}_rank{rank}.glb"))
            glb_mb = os.path.getsize(glb) / 1e6
            scene = Scene().load_gltf(glb)
            scene_source = f"binary glTF 2.0, {glb_mb:.0f} MB, written by host/gltf_export.cpp and imported by host/gltf.cpp ({time.time() - t_io:.1f} s)"
    animated = args.workload == "spheres10k"
    if animated:
        scene.build("spheres", 100, 100, 0.28)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    # Frames per launch and frames in flight.  Default: B = 8 independent frames traced per rfw_hip_render_batch call (one launch per
    # stage over the paths of all 8; with a sharded frame ONE all-gather per batch), 3 such batches in flight.  Measured on one GPU:
    # 5350-5400 Mrays/s, against 5160 with `--batch 1 --frames-in-flight 8` (eight frame slots, one render() -- the reference's own call
    # -- per frame).  A sharded frame needs the batches more: a rank's kernels cover 1/N of a frame and their time is the slowest
    # wavefront's; --emulate-shard 8, ms per frame on rank 0's shard: one frame at a time 0.42, 12 in flight 0.116, 3 batches of 8 in
    # flight 0.098.  C3 changes the scene every frame (a batch shares one scene): one render() per frame, 8 (12 for N > 2) in flight.
    B = args.batch if args.batch > 0 else 8
    if animated:
        B = 1
    F = args.frames_in_flight if args.frames_in_flight > 0 else (3 if B > 1 else 8 if world <= 2 else 12)
    # HOW the frames in flight are held.  One GPU: ONE renderer instance with F frame slots (rfw_hip_options.frames_in_flight: one scene
    # in HBM; path state, stream and TLAS per slot, so C3's per-frame instance updates pipeline too).  Sharded frame (N > 1): F instances
    # used round-robin, each with its own scene copy, because every frame in flight then needs its own all-gather buffers.
    use_slots = world == 1 and not args.emulate_shard and F > 1 and os.environ.get("RFW_BENCH_INSTANCES") is None
    n_inst = 1 if use_slots else F
    bes, streams, gathers = [], [], []
    t0 = time.time()
    for f in range(n_inst):
        be = HipBackend.init(w, h, 1.0, device=dev, max_path_length=args.max_path_length, rank=rank if not args.emulate_shard else 0,
                             world=world if not args.emulate_shard else args.emulate_shard,
                             streams=int(os.environ.get("RFW_STREAMS", "0")), frames_in_flight=F if use_slots else 0, max_batch=B)
        # each instance launches on its own HIP stream; torch wraps THAT stream (no second stream is created: HIP deals streams to
        # a few hardware queues in creation order, and two instances whose streams share a queue would serialise), so RCCL's
        # all-gather is ordered against the kernels and the HIP events that time them are recorded on the launch stream
        st = torch.cuda.ExternalStream(be.stream_handle(), device=dev)
        for key in ("sah_max_leaf", "sah_trav_cost"):  # builder experiments
            if os.environ.get("RFW_" + key.upper()):
                be.set_option(key, float(os.environ["RFW_" + key.upper()]))
        scene.mark_all_changed()
        scene.sync(be)
        g = None
        if world > 1 or args.emulate_shard:
            nslab = be.shard_info()["slab_floats"]
            wn = world if not args.emulate_shard else args.emulate_shard
            send = torch.zeros(B * nslab, dtype=torch.float32, device="cuda")           # this rank's tiles of B frames (written by render())
            g = (send, torch.zeros(wn * B * nslab, dtype=torch.float32, device="cuda"), nslab, wn)  # (send buffer, all ranks' slabs)
            be.set_slab_output(send.data_ptr())
        bes.append(be); streams.append(st); gathers.append(g)
    sync_s = (time.time() - t0) / n_inst
    be = bes[0]
    sstats = be.scene_stats()
    torch.cuda.synchronize()

    frame_no = [0]
    sync_ms = [0.0]
    step_no = [0]

    host_ring = {b_: [[b_.host_frame(presented=args.readback == "presented") for _ in range(2 * max(F if use_slots else 1, B))], 0] for b_ in bes} if args.readback else {}
    pending = [0]
    last_issue = [None]

    def step():
        """One frame.  With --batch B the frame is queued and every B-th call traces the B queued frames in one render_batch()."""
        if B > 1:
            pending[0] += 1
            if pending[0] == B:
                flush()
            return
        issue(1)

    def flush():
        if pending[0]:
            issue(pending[0])
            pending[0] = 0

    def issue(nf):
        k = step_no[0] % n_inst
        step_no[0] += 1
        b, g = bes[k], gathers[k]
        last_issue[0] = (b, nf)
        with (torch.cuda.stream(streams[k]) if world > 1 else contextlib.nullcontext()):  # the library already launches on streams[k]
            if animated:  # C3: every instance moves every frame (examples/animated/src/main.rs:197-219) -> set_3d_instances + synchronize
                t_s = time.perf_counter()
                scene.animate(frame_no[0] / 60.0)
                frame_no[0] += 1
                scene.sync(b)
                sync_ms[0] += (time.perf_counter() - t_s) * 1e3
            if B > 1:
                b.render_batch([view] * nf)  # nf independent new images (here of the same view, like the reset + render below)
            else:
                b.reset_accumulation()
                b.render(view)
            if world > 1:
                send, recv, nslab, wn = g
                send, recv = send[:nf * nslab], recv[:wn * nf * nslab]  # slab = [frame][tile pixels]; gathered = [rank][frame][tile pixels]
                if dist_backend == "nccl":
                    dist.all_gather_into_tensor(recv, send)  # the ONE collective per frame / per batch (RCCL over xGMI)
                else:
                    host = torch.empty(recv.shape, dtype=recv.dtype)
                    dist.all_gather_into_tensor(host, send.cpu())
                    recv.copy_(host)
                b.assemble_batch(recv.data_ptr(), nf)
            elif args.emulate_shard:
                b.assemble_batch(g[1].data_ptr(), nf)  # the de-tiling a rank would do after the all-gather
            if args.readback:  # every finished frame to (pinned) host memory, queued behind its kernels: the DMA overlaps the next frames' tracing
                for f_ in range(nf):
                    ring = host_ring[b]
                    dst = ring[0][ring[1] % len(ring[0])]
                    if ring[1] >= len(ring[0]):  # the ring hands this buffer out again: its copy (2 F frames ago) must have landed
                        b.wait_downloads(dst)
                    b.download_frame(dst, frame=f_)
                    ring[1] += 1

    # algorithmic bytes per ray from the traversal's own visit counters (one instrumented frame, untimed)
    be.set_option("count_traversal", 1)
    issue(1)
    cs = be.frame_stats()
    be.set_option("count_traversal", 0)
    rays_local = cs["primary_rays"] + cs["shadow_rays"] + cs["extension_rays"]

    for _ in range(args.warmup):
        step()
    flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = {"ms_trace_primary": 0.0, "ms_trace_shadow": 0.0, "ms_shade": 0.0, "ms_total": 0.0}
    for b in bes:
        b.drain_timing()
    # per-kernel HIP events inside the timed region only when they mean something: with frames in flight the kernels of different
    # frames overlap and the per-kernel roofline comes from the isolated pass below, so the ~14 event records per frame are skipped
    events_in_timed_region = F == 1 and B == 1
    for b in bes:
        b.set_option("timing", 1 if events_in_timed_region else 0)
    timed_frames = 0
    sync_ms[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        if events_in_timed_region and ((i + 1) % (24 * F) == 0 or i + 1 == args.steps):
            # per-kernel HIP-event durations, recorded inside render() on the launch stream for EVERY timed frame and
            # read back in batches (one stream sync per 24 frames per instance instead of one per frame)
            for b in bes:
                ms, n = b.drain_timing()
                timed_frames += n
                for k in kernel_ms:
                    kernel_ms[k] += ms[k]
    flush()  # a last, shorter batch when --steps is not a multiple of --batch: exactly K frames are timed
    if args.readback:
        for b in bes:
            b.wait_downloads()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # Sharded frame: is what the all-gather + assemble left on this rank the frame one GPU renders alone?  (after the timed region;
    # static scenes only: C3's last frame depends on the animation clock)
    shard_check = None
    if world > 1 and rank == 0 and not animated and last_issue[0] is not None:
        lb, lnf = last_issue[0]
        whole = HipBackend.init(w, h, 1.0, device=dev, max_path_length=args.max_path_length)
        scene.mark_all_changed()
        scene.sync(whole)
        whole.render(view)
        ref = whole.accumulator().view(np.uint32)
        shard_check = all(np.array_equal(lb.accumulator_at(f).view(np.uint32), ref) for f in range(lnf))
        whole.close()
    # With several frames in flight the kernels of different frames share the machine, so their HIP-event spans in the timed
    # region are not launch durations.  The per-kernel roofline therefore comes from a short pass AFTER the timed region that
    # renders one frame at a time on instance 0 (the same thing `--frames-in-flight 1` times, and what a rocprofv3 kernel trace
    # of that command shows); the timed region's own aggregate (all kernels' bytes / ms_per_step) is reported next to it.
    iso_ms, iso_frames = None, 0
    if F > 1 or B > 1:
        iso_ms = {k: 0.0 for k in kernel_ms}
        bes[0].set_option("timing", 1)
        bes[0].drain_timing()
        for i in range(min(48, max(args.steps, 1))):
            bes[0].reset_accumulation()
            bes[0].render(view)
            if use_slots:
                bes[0].device_synchronize()  # one frame at a time although the instance would pipeline them over its slots
            if (i + 1) % 24 == 0:
                ms_, n_ = bes[0].drain_timing()
                iso_frames += n_
                for k in iso_ms:
                    iso_ms[k] += ms_[k]
        ms_, n_ = bes[0].drain_timing()
        iso_frames += n_
        for k in iso_ms:
            iso_ms[k] += ms_[k]
        torch.cuda.synchronize()
    # host cost of the per-frame scene update (animate + set_3d_instances + synchronize) without back-pressure: inside the timed
    # region a host that runs ahead of the GPU spends most of synchronize() waiting for a staging block, which is idle time
    host_sync_ms = None
    if animated:
        acc_t = 0.0
        for i in range(20):
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            scene.animate(frame_no[0] / 60.0)
            frame_no[0] += 1
            scene.sync(bes[0])
            acc_t += time.perf_counter() - t_s
            bes[0].render(view)
        torch.cuda.synchronize()
        host_sync_ms = acc_t / 20 * 1e3
    if world > 1:
        t = torch.tensor([elapsed, float(rays_local)], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        elapsed, rays_total = float(tmax[0]), float(t[1])
    else:
        rays_total = float(rays_local)

    # this job's own HBM roofline: a float4 device copy, measured after the timed region
    try:
        bw_measured = bes[0].bandwidth_probe(1 << 30, 20) if rank == 0 else None
    except Exception:
        bw_measured = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = rays_total / (elapsed / args.steps) / 1e6
        # roofline: ALGORITHMIC bytes each kernel moves per launch (DESIGN.md §"algorithmic bytes") / its mean HIP-event duration
        node_b, tri_b = sstats["node_bytes"], sstats["tri_bytes"]
        nf = max(timed_frames, 1)
        n_prim, n_shad = cs["primary_rays"], cs["shadow_rays"]
        alg = {
            # nodes x 128 B + triangles x 48 B + instance records x 64 B + 48 B written per ray (origin, direction, hit) + 16 B accumulator clear
            "k_primary": cs["nodes_visited"][0] * node_b + cs["tris_tested"][0] * tri_b + cs["instances_entered"][0] * 64 + n_prim * (48 + 16),
            # + 32 B queue entry read per ray, 16 B contribution read and 32 B accumulator read-modify-write per unoccluded ray (counted for all)
            "k_shadow": cs["nodes_visited"][2] * node_b + cs["tris_tested"][2] * tri_b + cs["instances_entered"][2] * 64 + n_shad * (32 + 16 + 32),
            # per path: hit 16 + ray 32 read; per hit: RTTriangle 176 + material 96 + normal matrix 48; shadow-queue push 48 per shadow ray
            "k_shade": n_prim * (16 + 32) + n_prim * (176 + 96 + 48) + n_shad * 48,
        }
        # a frame is split into `sub` sub-shards traced on separate streams: each kernel is launched `sub` times per frame; the
        # HIP-event durations below are per-frame SUMS over those launches, so bytes-per-frame / sum-of-durations is exactly
        # (bytes per launch) / (mean launch duration)
        sub = max(cs.get("substreams", 1), 1)
        ms_timed = {"k_primary": kernel_ms["ms_trace_primary"] / nf, "k_shadow": kernel_ms["ms_trace_shadow"] / nf, "k_shade": kernel_ms["ms_shade"] / nf}
        if iso_ms is not None and iso_frames > 0:
            ms = {"k_primary": iso_ms["ms_trace_primary"] / iso_frames, "k_shadow": iso_ms["ms_trace_shadow"] / iso_frames, "k_shade": iso_ms["ms_shade"] / iso_frames}
            measured = f"{iso_frames} frames rendered one at a time after the timed region (kernels of the {F * B} frames in flight overlap inside it)"
        else:
            ms = ms_timed
            measured = f"all {nf} frames of the timed region"
        dom = max(ms, key=lambda k: ms[k])
        gbs = {k: (alg[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0) for k in alg}
        achieved = gbs[dom]
        traffic = pmc_traffic(dom)
        rays_local_f = max(rays_local, 1)
        out = {
            "metric": "Mrays/s (primary+shadow, 1spp)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic atrium ({'glTF scene' if not args.procedural else 'procedural'}), {sstats['triangles']} triangles in {sstats['instances']} instance(s), {w}x{h}, 1 spp, "
                                   + ("primary+shadow (max path length 1)" if args.max_path_length == 1 else f"path traced, max path length {args.max_path_length}, NEE")
                                   + (", every instance moved and the TLAS rebuilt on the device every frame" if animated else ", static scene") + ", BVH4",
                       "scene_source": scene_source, "rays_per_frame": int(rays_total), "frames_in_flight": F * B, "batches_in_flight": F, "frames_per_batch": B, "frames_in_flight_held_by": "frame slots of one instance (one scene copy)" if use_slots else (f"{n_inst} renderer instances" if n_inst > 1 else "-"), "readback_every_frame": args.readback or False, "sharded_frame_equals_single_gpu_frame": shard_check, "tile_shard": "64x64 round-robin" if world > 1 else "none",
                       "bvh": {"blas_nodes": sstats["blas_nodes"], "node_bytes": node_b, "tri_bytes": tri_b, "build_ms": round(sstats["ms_blas_build"], 1)},
                       "synchronize_s": round(sync_s, 2), "max_path_length": args.max_path_length,
                       "instances": sstats["instances"], "tlas_nodes": sstats["tlas_nodes"],
                       "per_frame_synchronize_ms": round(host_sync_ms, 3) if animated else None,
                       "per_frame_synchronize_wall_ms_in_timed_region": round(sync_ms[0] / args.steps, 3) if animated else None},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "peak_measured": round(bw_measured, 1) if bw_measured else None,   # device float4 copy in this job, read + write
                         "frac_of_measured": round(achieved / bw_measured, 4) if bw_measured else None,
                         "algorithmic_bytes_per_launch": int(alg[dom] / sub), "avg_launch_ms": round(ms[dom] / sub, 4), "launches_per_frame": sub,
                         "per_kernel": {k: {"ms_sum_per_frame": round(ms[k], 4), "alg_GBps": round(gbs[k], 1), "frac": round(gbs[k] / HBM_PEAK_GBS, 4)} for k in alg},
                         "nodes_per_ray": {"primary": round(cs["nodes_visited"][0] / max(n_prim, 1), 2), "shadow": round(cs["nodes_visited"][2] / max(n_shad, 1), 2)},
                         "tris_per_ray": {"primary": round(cs["tris_tested"][0] / max(n_prim, 1), 2), "shadow": round(cs["tris_tested"][2] / max(n_shad, 1), 2)},
                         # SIMD efficiency of the traversal, from the instrumented frame: active lanes / 64 per execution of the node test
                         # and of the triangle test, and what lanes that finished before their wavefront cost (nodes / (64 x max per wave))
                         "lane_utilisation": {name: {"node_test": round(cs["nodes_visited"][k] / max(64 * cs["node_test_executions"][k], 1), 3),
                                                     "triangle_test": round(cs["tris_tested"][k] / max(64 * cs["tri_test_executions"][k], 1), 3),
                                                     "finished_lanes_bound": round(cs["nodes_visited"][k] / max(64 * cs["wave_max_nodes"][k], 1), 3)}
                                              for k, name in ((0, "primary"), (1, "extension"), (2, "shadow")) if cs["node_test_executions"][k]},
                         "frame_ms_events": round((iso_ms["ms_total"] / iso_frames) if (iso_ms is not None and iso_frames > 0) else kernel_ms["ms_total"] / nf, 4),
                         "measured": measured,
                         # what the path is closest to: vector-instruction issue.  gfx950 runs FP32 vector instructions at 32 lanes
                         # per SIMD and clock, so the peak is 1024 SIMDs x 2.4 GHz / 2 cycles = 1228.8 G wave64 instructions/s (= the 157 TFLOP/s
                         # FP32 vector peak); tools/probes/valu_issue_probe.hip measures 880-1000 G/s for independent FMAs and 517 G/s when every
                         # instruction depends on the one before.  Instructions per frame: the committed PMC profile of THIS workload at max path
                         # length 1 (null otherwise)
                         "valu_issue": (lambda v: {"wave_instructions_per_frame": v, "peak_per_s": 1228.8e9, "measured_independent_fma_per_s": 1.0e12,
                                                   "measured_dependent_chain_per_s": 0.517e12,
                                                   "frac": round(v / (ms_step * 1e-3) / 1228.8e9, 4)} if v else None)(
                             pmc_valu_per_frame() if (args.workload == "atrium1m" and args.max_path_length == 1 and world == 1 and not args.emulate_shard) else None),
                         # the timed region as a whole: every kernel's algorithmic bytes of one frame over the wall time per frame
                         "timed_region": {"frames_in_flight": F * B, "algorithmic_bytes_per_frame": int(sum(alg.values())),
                                          "achieved": round(sum(alg.values()) / (ms_step * 1e-3) / 1e9, 1),
                                          "frac": round(sum(alg.values()) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "per_kernel_events": events_in_timed_region}},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, view, w, h, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for b in bes:
        b.close()


def pmc_valu_per_frame():
    """Wave64 VALU instructions one frame of the headline workload issues, from the committed rocprofv3 PMC summary
    (profiles/*_pmc_valu.json: SQ_INSTS_VALU per launch of k_primary, k_shade, k_shadow, k_assemble), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_valu.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"]
        total = 0
        # the non-counting, single-frame instantiations of the frame's four kernels (names as the profiler prints them, older sets included)
        for names in (("k_primary<false>",), ("k_shade<false>", "k_shade"), ("k_shadow<false, false>", "k_shadow<false,false>", "k_shadow<false>"),
                      ("k_assemble<false, false>", "k_assemble<false>", "k_assemble")):
            total += k[next(n for n in names if n in k)]["SQ_INSTS_VALU"]
        return int(total)
    except Exception:
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this workload (profiles/*_pmc.json,
    written by tools/summarize_profile.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes; gfx950 correction applied).
    Counters cannot be read from inside this process, so this is the latest committed measurement, or null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"]
        for name in (kernel + "<false, false>", kernel + "<false>", kernel):  # the non-counting single-frame instantiation
            if name in k:
                return k[name]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None
    return None


def cpu_baseline(scene, view, w, h, budget_s):
    """The oracle (CPU restatement of the reference's rtbvh/MBVH path) timed on this host's cores on a bounded
    sample of the same workload: the same scene and camera at reduced resolution, 1 spp primary + shadow."""
    from oracle.bindings import Oracle
    cores = os.cpu_count() or 1
    sw, sh = 480, 270
    orc = Oracle(sw, sh, threads=cores, max_path_length=1)
    scene.mark_all_changed()
    t0 = time.time()
    scene.sync(orc)
    build_s = time.time() - t0
    v = scene.view(sw, sh)
    orc.render(v)  # warm
    s0 = orc.stats()
    n, t0 = 0, time.perf_counter()
    while True:
        orc.reset()
        orc.render(v)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    s = orc.stats()
    rays = s["primary"] + s["shadow"]
    return {"value": round(rays * n / el / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"same scene and camera at {sw}x{sh}, 1 spp primary+shadow, {n} frames, {cores} threads; BVH build {build_s:.1f}s excluded"}


if __name__ == "__main__":
    main()
