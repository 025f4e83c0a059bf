#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + shadow, 1 spp) of the hot path on N MI355X (contract in the task statement).

A step = one frame: one 1920x1080 image of the ~1M-triangle synthetic atrium (the configuration the metric's target is quoted on,
BASELINE.json north_star; SURVEY.md §8d C4 geometry, primary + shadow rays), every input resident in HBM.

N = 1: `value` is what a client of the reference's plugin trait gets — ONE `Backend::render` call per frame and nothing else
(rfw_hip_render; 12 frame slots inside the instance keep frames in flight because every frame has its own camera view, a slow dolly,
so each call starts a new image).  Two further modes are measured after the timed region and reported beside it (`config.modes`):
strictly one frame at a time, and frames traced 8 per launch through the rfw_hip_render_batch extension.  The first and the last
frame of the timed region are read back and compared, bit for bit, with the CPU oracle's frames of the same views at the same
resolution (`config.timed_frame_equals_oracle`); the oracle's time for those frames is the `cpu_baseline`.

N > 1: the frame is sharded by 64x64 tiles across ranks; `value` is north_star's protocol — one render() and ONE all-gather of the
framebuffer per frame (RCCL over xGMI; strong scaling: the frame is fixed), 8 (N = 2) or 12 renderer instances in flight.  Batches of 8
frames traced per launch with one exchange per batch (the rfw_hip_render_batch extension) are measured after the timed region and reported
as a row of config.modes."""
import argparse
import contextlib
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Every frame slot / renderer instance launches on its own HIP stream, and the runtime deals streams to hardware queues (4 by
# default): two streams on one queue serialise.  Ask for enough queues BEFORE the HIP runtime starts (it reads this once).
# (main() chooses once it knows the configuration: 16 hardware queues, 24 where more than 16 streams are in use — N > 2, or C3's 20 frame slots)
QUEUES_SET_BY_CALLER = "GPU_MAX_HW_QUEUES" in os.environ

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured for a float4 copy)
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md §L2: ~34.5 TB/s aggregate
L1_PEAK_GBS = 256 * 64 * 2.4  # 64 B per clock and CU out of the vector L1 / TA return path x 256 CUs x 2.4 GHz = 39.3 TB/s
ISSUE_MEASURED = {}          # filled on rank 0 by rfw_hip_issue_probe after the timed region: G wave64 instructions/s this device sustains in this job
VALU_PEAK_GIPS = 1228.8      # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 FP32 instruction (= the 157 TFLOP/s vector peak)
CLOCK_GHZ = 2.4              # peak engine clock (the busy-cycle fractions below are conservative if the chip clocks lower under load)
N_VIEWS = 16                 # distinct camera views the frames cycle through


def kernel_source_hash():
    """sha1 over the sources the device code is built from (rfw-rs_amd/csrc: *.hip, *.h, *.inc, *.cpp, Makefile), in name order.  Counter
    summaries under profiles/ carry the hash of the sources they were measured on (tools/summarize_profile.py); instruction and request
    counts are properties of (scene, view, kernel code), so they may only be divided by THIS run's durations when the code is the same."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "rfw-rs_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def config_key(args):
    """The configuration of BASELINE.json this run is (the tag of its counter summaries under profiles/), or None."""
    if args.width != 1920 or args.height != 1080 or args.identical_frames or args.emulate_shard:
        return None
    if args.workload == "atrium1m":
        key = {1: "c4", 3: "c4path"}.get(args.max_path_length)
        # (tools/measure.sh's c4pathS pass: streaming forced through the environment, so that the one-frame-at-a-time counter runs see the kernels
        # the bounces run with frames in flight — its bench line reads ITS OWN counter set, whose kernels are k_extend_stream / k_shadow_stream)
        return "c4pathS" if key == "c4path" and os.environ.get("RFW_STREAM_RUN") else key
    if args.max_path_length != 1:
        return None
    return {"atrium262k": "c2", "spheres10k": "c3", "atrium32m": "c32m"}.get(args.workload)


def dolly_views(base, n, step):
    """n camera views: `base` translated k * step along its viewing direction (pos and the virtual screen's corner move together)."""
    from rfw_rs_amd import pod
    out = []
    for k in range(n):
        v = pod.CameraView3D.from_buffer_copy(base)
        for a in "xyz":
            d = getattr(base.direction, a) * step * k
            setattr(v.pos, a, getattr(base.pos, a) + d)
            setattr(v.p1, a, getattr(base.p1, a) + d)
        out.append(v)
    return out


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N` without a launcher around it: start N ranks of this same script as CHILD processes, one per GPU, relay
    rank 0's JSON line, and fail if any rank fails.  Runs BEFORE anything in this process touches the GPU (no torch.cuda / HIP call has
    happened, torch is not even imported), and never replaces a process (no os.exec*): the parent only waits.  The children see
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as `python -m torch.distributed.run` would set them."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RFW_BENCH_LAUNCHED_BY=str(os.getpid()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if not QUEUES_SET_BY_CALLER:
            env.pop("GPU_MAX_HW_QUEUES", None)  # (this process chose it for ONE rank: a child chooses for its own WORLD_SIZE)
        # rank 0's stdout is the job's stdout (the ONE JSON line); what the other ranks print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else 2))  # (fd 2: this process's stderr, inherited as is)
    lines = []

    def relay():
        for raw in procs[0].stdout:
            line = raw.decode(errors="replace")
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in alive:  # exactly the processes started above, by handle
                    procs[o].terminate()
        time.sleep(0.05)
    t.join(timeout=10)
    if rc == 0 and not any(l.startswith("{") for l in lines):
        print("bench.py: every rank exited 0 but rank 0 printed no result line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher self-test (runs without a GPU): the ranks only meet (gloo), rank 0 prints who it saw; no scene, no rendering")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="atrium1m", choices=["atrium1m", "atrium262k", "cornell", "spheres10k", "atrium32m"],
                    help="atrium32m = the same atrium at 33.5 M triangles (2 GB of nodes, 1.6 GB of packets: far outside every cache; use --procedural); "
                         "atrium1m = headline (C4 geometry, primary+shadow); atrium262k = C2; spheres10k = C3 (atrium262k + 10 000 "
                         "animated icosphere instances, synchronize() every frame); cornell = C1 geometry")
    ap.add_argument("--max-path-length", type=int, default=1, help="1 = primary+shadow (the metric); 3 = the reference's path tracer (C4)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frame slots (N = 1) / renderer instances (N > 1) used round-robin; 1 = strictly one frame at a time")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames traced per rfw_hip_render_batch call.  Default 1 at every N: one trait call per frame and, at N > 1, ONE exchange of the "
                         "framebuffer per frame (north_star's protocol: that is the `value`); at N > 1 batches of 8 — one exchange per batch — are measured "
                         "after the timed region and reported as a row of config.modes")
    ap.add_argument("--procedural", action="store_true",
                    help="hand the generated scene to the backend directly; default: write it as a binary glTF 2.0 file (host/gltf_export.cpp) and "
                         "run on what the glTF importer (host/gltf.cpp) reads back — the configurations of BASELINE.json are glTF scenes")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (CPU baseline + check of the timed frames)")
    ap.add_argument("--no-modes", action="store_true", help="skip the secondary modes (one frame at a time / batches) after the timed region")
    ap.add_argument("--separate-spheres", action="store_true", help="atrium1m: C4's 64 displaced icospheres as 64 meshes with one instance each (65 meshes, SURVEY §8d literally) instead of one baked mesh")
    ap.add_argument("--identical-frames", action="store_true", help="every frame the same view (round 1's configuration; frames then share every cache line)")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="single-GPU study of multi-GPU scaling: render only rank 0's tiles of an N-way tile shard (and de-tile a stand-in "
                         "gathered buffer), without any collective; the value then counts this shard's rays only")
    ap.add_argument("--collective", default="torch", choices=["torch", "native", "p2p"],
                    help="N > 1: who exchanges the tiles — torch.distributed's all-gather (RCCL through PyTorch), the library's own (rfw_hip_comm_*: librccl on the "
                         "instance's stream), or no collective at all: p2p = every rank stores its tiles into the destinations' buffers over xGMI (rfw_hip_p2p_*)")
    ap.add_argument("--gather-format", default="bgra8", choices=["f32", "f16", "bgra8"],
                    help="N > 1: what a rank's tiles travel as — the accumulator's RGB floats (12 B per pixel), the finished frame as halves (6 B) or the "
                         "presented B, G, R, A bytes (4 B: the swap-chain image Backend::render ends with; default).  Accumulation stays in f32 on the "
                         "rank that owns the tiles in every format")
    ap.add_argument("--present-rank", type=int, default=0, help="N > 1: the rank that de-tiles every gathered frame at once (-1: all ranks do)")
    ap.add_argument("--readback", nargs="?", const="float", default=None, choices=["float", "presented"],
                    help="copy every finished frame to (pinned) host memory inside the timed region (the PCIe-inclusive rate of DESIGN.md; never the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--mode-frames", type=int, default=48, help="frames per secondary mode")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="the headline run (atrium1m, max path length 1, one GPU) also runs BASELINE.json's other single-GPU configurations — C2, C3 and the "
                         "path-traced C4 — as short child runs of this script, each with its own oracle check, and reports them in config.modes; this skips them")
    ap.add_argument("--baseline-steps", type=int, default=200, help="timed frames of each of those child runs")
    args = ap.parse_args()

    # ---- who runs the ranks.  `--gpus N` is the number of ranks, whoever starts them:
    #   * under a launcher (python -m torch.distributed.run: WORLD_SIZE = N in the environment) this process IS one of the N ranks;
    #   * plain `python3 bench.py --gpus N` starts the N ranks itself, as child processes, before it touches the GPU (launch_ranks);
    #   * `--gpus 1` is ONE rank whatever a stray WORLD_SIZE in the environment says.
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and env_world == 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks: refusing to report a number for a job of another size")
    world = args.gpus
    if not QUEUES_SET_BY_CALLER:  # (before anything starts the HIP runtime)
        os.environ["GPU_MAX_HW_QUEUES"] = "24" if (world > 2 or args.workload == "spheres10k") else "16"
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    # RFW_BENCH_DIST_BACKEND=gloo is a TEST HOOK: it lets two ranks share one GPU (RCCL refuses duplicate devices) so the
    # N > 1 code path can be exercised on a 1-GPU box; slabs are then staged through host memory.  Default: nccl (= RCCL).
    dist_backend = os.environ.get("RFW_BENCH_DIST_BACKEND", "nccl")

    # BASELINE.json's other single-GPU configurations as child runs (baseline_configs), BEFORE this process starts the HIP runtime: a second
    # process beside one that holds its 16 hardware queues — even idle ones — oversubscribes the device's queues and runs 5-13 % slower
    # (measured: C3 as a child of an initialised parent 5690, alone 6568 Mrays/s on the same box)
    baseline_rows = {}
    if (world == 1 and not args.emulate_shard and not args.no_baseline_configs and not args.no_modes and not args.rendezvous_only and args.workload == "atrium1m"
            and args.max_path_length == 1 and (args.width, args.height) == (1920, 1080) and not args.identical_frames and not args.readback):
        baseline_rows = baseline_configs(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.rendezvous_only:
        # the launcher's self-test: the ranks meet over gloo and say who they are; nothing else of the bench runs (no GPU needed)
        seen = [None]
        if world > 1:
            dist.init_process_group("gloo")
            seen = [None] * dist.get_world_size()
            dist.all_gather_object(seen, {"rank": dist.get_rank(), "pid": os.getpid(), "launched_by": os.environ.get("RFW_BENCH_LAUNCHED_BY")})
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = [{"rank": 0, "pid": os.getpid(), "launched_by": os.environ.get("RFW_BENCH_LAUNCHED_BY")}]
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": len(seen), "ranks_seen": len(seen), "ranks": seen}), flush=True)
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    n_dev = torch.cuda.device_count()
    if world > n_dev and dist_backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {world} but this node shows {n_dev} device(s) (RCCL takes one rank per device; RFW_BENCH_DIST_BACKEND=gloo is the "
                         "test hook that lets ranks share a device)")
    dev = (local_rank % n_dev) if world > 1 else 0
    torch.cuda.set_device(dev)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(dist_backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {world}")

    from rfw_rs_amd import HipBackend, Scene

    w, h = args.width, args.height
    tris = {"atrium1m": 1048576, "atrium262k": 262267, "cornell": 0, "spheres10k": 262267, "atrium32m": 33554432}[args.workload]
    scene = Scene().build("cornell") if args.workload == "cornell" else Scene().build("atrium", tris, int(os.environ.get("RFW_SPHERE_MESHES", "1" if args.separate_spheres else "0")), 0.0, 0xC0FFEE)
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
    base_view = scene.view(w, h)
    # every frame its own view: a dolly of 5 mm per frame along the viewing direction, 16 views cycled.  Frames in flight then differ in
    # every ray (round 1 traced 24 bit-identical frames), and one render() per view is a new image each time — which is what lets the
    # trait's own call pipeline over the frame slots.  C3 keeps one view: there the scene changes every frame.
    views = [base_view] * N_VIEWS if (args.identical_frames or animated) else dolly_views(base_view, N_VIEWS, 0.005)

    single = world == 1 and not args.emulate_shard
    B = args.batch if args.batch > 0 else 1
    if animated:
        B = 1
    # (C3: a frame's chain is longer — instance upload, TLAS rebuild, then the trace: measured 5000 / 4910 / 5080 / 5160 Mrays/s with 4 / 8 / 12 /
    # 16 slots on one box in round 3; round 6, interleaved on one box, three runs each: 16 slots 6919-7071, 20 slots with 24 hardware queues
    # 7099-7276; beyond 22 slots the streams outnumber the queues the hardware has and the rate collapses)
    # (static scenes, round 4: 8 / 12 / 16 slots give 7050 / 7096 / 7118 Mrays/s over 400 frames and 6560 / 6680 / 6730 over the driver's 20: 12 it is)
    F = args.frames_in_flight if args.frames_in_flight > 0 else (3 if B > 1 else 20 if (animated and world == 1) else 12 if world == 1 else 8 if world <= 2 else 12)
    # HOW the frames in flight are held.  One GPU: ONE renderer instance with F frame slots (rfw_hip_options.frames_in_flight: one scene
    # in HBM; path state, stream and TLAS per slot, so C3's per-frame instance updates pipeline too).  Sharded frame (N > 1): F instances
    # used round-robin, each with its own scene copy, because every frame in flight then needs its own all-gather buffers.
    p2p = args.collective == "p2p" and world > 1
    native = (args.collective == "native" and world > 1 and dist_backend == "nccl") or p2p  # the exchange happens inside render()
    # (with the library's own exchange the frame slots of ONE instance share it: one scene copy per rank at N > 1 too)
    use_slots = (single or native) and F > 1 and os.environ.get("RFW_BENCH_INSTANCES") is None
    gather_format = {"f32": 0, "f16": 1, "bgra8": 2}[args.gather_format] if (world > 1 or args.emulate_shard) else 0

    def make_instances(n_slots_or_inst, batch, slots):
        n_inst = 1 if slots else n_slots_or_inst
        out = []
        for f in range(n_inst):
            be = HipBackend.init(w, h, 1.0, device=dev, max_path_length=args.max_path_length, rank=rank if not args.emulate_shard else 0,
                                 world=world if not args.emulate_shard else args.emulate_shard,
                                 streams=int(os.environ.get("RFW_STREAMS", "0")), builder=int(os.environ.get("RFW_BUILDER", "0")), frames_in_flight=n_slots_or_inst if slots else 0, max_batch=batch)
            # each instance launches on its own HIP stream; torch wraps THAT stream (no second stream is created), so RCCL's
            # all-gather is ordered against the kernels
            st = torch.cuda.ExternalStream(be.stream_handle(), device=dev)
            for key in ("sah_max_leaf", "sah_trav_cost", "sort_extension_rays", "stream_run", "stream_refill", "stream_leaf_gate", "shade_group", "tlas_fused"):  # A/B experiments: RFW_<OPTION>=value
                if os.environ.get("RFW_" + key.upper()):
                    be.set_option(key, float(os.environ["RFW_" + key.upper()]))
            if world > 1 or args.emulate_shard:
                be.set_option("gather_format", gather_format)
                be.set_option("present_rank", args.present_rank)
            scene.mark_all_changed()
            scene.sync(be)
            g = None
            if (world > 1 and not native) or args.emulate_shard:
                nslab = be.shard_info()["slab_floats"]  # 4-byte words of one frame's slab in the gather format
                wn = world if not args.emulate_shard else args.emulate_shard
                send = torch.zeros(batch * nslab, dtype=torch.float32, device="cuda")           # this rank's tiles of `batch` frames (written by render())
                g = (send, torch.zeros(wn * batch * nslab, dtype=torch.float32, device="cuda"), nslab, wn)  # (send buffer, all ranks' slabs)
                be.set_slab_output(send.data_ptr())
            out.append((be, st, g))
        return out

    t0 = time.time()
    inst = make_instances(F, B, use_slots)
    sync_s = (time.time() - t0) / len(inst)
    def connect(insts):
        if p2p:
            # every rank's 256-byte handle to every rank, once; after that the ranks only meet in their peers' flag words
            for be_, _, _ in insts:
                hds = [None] * world
                dist.all_gather_object(hds, be_.p2p_export())
                be_.p2p_connect(hds)
            dist.barrier()
        elif native:
            # the communicators live inside the library (librccl; one per instance, i.e. per frame / batch in flight): rank 0's unique ids
            # travel through torch's store once, after that torch.distributed is only used for the barrier around the timed region
            uids = [[HipBackend.comm_unique_id() for _ in insts] if rank == 0 else None]
            dist.broadcast_object_list(uids, src=0)
            for (be_, _, _), uid in zip(insts, uids[0]):
                be_.comm_init(uid, rank, world)

    connect(inst)
    bes = [i[0] for i in inst]
    be = bes[0]
    sstats = be.scene_stats()
    torch.cuda.synchronize()
    # the acceleration-structure build, warm: everything changed -> synchronize(), three times; the device's share by events
    build_report = None
    if rank == 0 and single and not animated:
        warm = []
        for _ in range(3):
            scene.mark_all_changed()
            t_b = time.perf_counter()
            scene.sync(be)
            be.device_synchronize()
            warm.append((time.perf_counter() - t_b) * 1e3)
        bs = be.scene_stats()
        up_ms, k_ms = bs["ms_blas_upload"], bs["ms_blas_kernels"]
        build_report = {
            "synchronize_warm_ms": [round(x, 2) for x in warm],
            "note": "synchronize() after every mesh changed: host copies of the meshes, host -> device copy of the triangles (176 B each), then on the device boxes, "
                    "binned-SAH builder, leaf-ordered packets, quantised nodes; the last two figures by HIP events on the instance's stream",
            "triangle_upload": {"ms": round(up_ms, 3), "bytes": bs["blas_upload_bytes"], "GBps": round(bs["blas_upload_bytes"] / max(up_ms, 1e-6) / 1e6, 1)},
            "device_kernels": {"ms": round(k_ms, 3), "algorithmic_bytes": bs["blas_kernel_bytes"],
                               "algorithmic_GBps": round(bs["blas_kernel_bytes"] / max(k_ms, 1e-6) / 1e6, 1),
                               "frac_of_hbm_peak": round(bs["blas_kernel_bytes"] / max(k_ms, 1e-6) / 1e6 / HBM_PEAK_GBS, 4),
                               "bound": "kernel launches and device-scope atomics on the upper levels (few nodes, each across thousands of workgroups), not bandwidth"},
        }

    frame_no = [0]
    sync_ms = [0.0]

    def animate_and_sync(b):
        # C3: every instance moves every frame (examples/animated/src/main.rs:197-219) -> set_3d_instances + synchronize
        t_s = time.perf_counter()
        scene.animate(frame_no[0] / 60.0)
        frame_no[0] += 1
        scene.sync(b)
        sync_ms[0] += (time.perf_counter() - t_s) * 1e3

    # ---- rays per view, counted by the traversal itself (one instrumented frame per view, untimed): exact ray totals for the timed
    # region and the algorithmic bytes of SURVEY §8(d)
    be.set_option("count_traversal", 1)
    per_view = []
    for k in range(N_VIEWS if not (animated or args.identical_frames) else 1):
        if animated:
            animate_and_sync(be)
        be.reset_accumulation()
        be.render(views[k])
        s = be.frame_stats()
        per_view.append(s)
    be.set_option("count_traversal", 0)
    be.reset_accumulation()
    cs = per_view[0]
    rays_of_view = [float(s["primary_rays"] + s["shadow_rays"] + s["extension_rays"]) for s in per_view]
    if world > 1:  # a rank counts its own shard: sum over ranks
        t = torch.tensor(rays_of_view, dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rays_of_view_total = [float(x) for x in t]
    else:
        rays_of_view_total = rays_of_view

    def view_rays(i):
        return rays_of_view_total[i % len(rays_of_view_total)]

    # ---- who rendered: one record per rank — the device it ran on and the rays of ITS shard of view 0 — gathered over the process group.
    # `n_gpus` in the result line is what this list shows, not what the command line asked for: a launcher detail that left N - 1 ranks
    # out would otherwise print a single-GPU number under `n_gpus: N`.
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "device": dev, "device_name": props.name,
          "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
          "rays_of_view_0": int(rays_of_view[0]), "tiles": be.shard_info()["tiles_local"]}
    ranks_report = [me]
    if world > 1:
        ranks_report = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_report, me)
    ranks_seen = len(ranks_report)
    ranks_rendered = sum(1 for r_ in ranks_report if r_ and r_["rays_of_view_0"] > 0)
    distinct_devices = len({(r_["device_uuid"] or r_["pci_bus_id"] or r_["device"]) for r_ in ranks_report})
    if ranks_seen != world or ranks_rendered != world:
        raise SystemExit(f"bench.py: --gpus {world}, but {ranks_seen} rank(s) met and {ranks_rendered} rendered a shard: refusing to print a line under n_gpus = {world}")
    if world > 1 and dist_backend == "nccl" and distinct_devices != world:
        raise SystemExit(f"bench.py: {world} ranks on {distinct_devices} distinct device(s): not a {world}-GPU job")

    # ---- one frame / one batch of frames
    host_ring = {id(b_): [[b_.host_frame(presented=args.readback == "presented") for _ in range(2 * max(F if use_slots else 1, B))], 0] for b_ in bes} if args.readback else {}
    state = {"issued": 0, "pending": [], "k": 0, "last": None}

    def issue(frames):
        """frames = list of view indices traced by ONE call (len 1: render(); more: render_batch())."""
        k = state["k"] % len(inst)
        state["k"] += 1
        b, st, g = inst[k]
        state["last"] = (b, list(frames))
        nf = len(frames)
        with (torch.cuda.stream(st) if (world > 1 and not native) else contextlib.nullcontext()):  # the library already launches on st
            if animated:
                animate_and_sync(b)
            if nf > 1:
                b.render_batch([views[i % N_VIEWS] for i in frames])
            elif args.identical_frames or animated or not use_slots:
                # (a repeated view would accumulate on its slot instead of starting a new image: reset first; C3's scene changed anyway)
                b.reset_accumulation()
                b.render(views[frames[0] % N_VIEWS])
            else:
                b.render(views[frames[0] % N_VIEWS])  # THE headline call: Backend::render, nothing else
            if world > 1 and native:
                pass  # render() / render_batch() gathered and assembled the frame(s) themselves: ncclAllGather on the instance's stream
            elif world > 1:
                send, recv, nslab, wn = g
                send, recv = send[:nf * nslab], recv[:wn * nf * nslab]  # slab = [frame][tile pixels]; gathered = [rank][frame][tile pixels]
                if dist_backend == "nccl":
                    dist.all_gather_into_tensor(recv, send)  # the ONE collective per batch (RCCL over xGMI)
                else:
                    host = torch.empty(recv.shape, dtype=recv.dtype)
                    dist.all_gather_into_tensor(host, send.cpu())
                    recv.copy_(host)
                b.assemble_batch(recv.data_ptr(), nf)
            elif args.emulate_shard:
                b.assemble_batch(g[1].data_ptr(), nf)  # the de-tiling a rank would do after the all-gather
            if args.readback:
                for f_ in range(nf):
                    ring = host_ring[id(b)]
                    dst = ring[0][ring[1] % len(ring[0])]
                    if ring[1] >= len(ring[0]):
                        b.wait_downloads(dst)
                    b.download_frame(dst, frame=f_)
                    ring[1] += 1

    def step(i):
        if B > 1:
            state["pending"].append(i)
            if len(state["pending"]) == B:
                flush()
        else:
            issue([i])

    def flush():
        if state["pending"]:
            issue(state["pending"])
            state["pending"] = []

    # frame 0 of the timed region is copied out (asynchronously, behind its kernels, into pinned memory) for the oracle check; the buffer is
    # allocated here, before the warm-up, so that nothing but the synchronisation sits between the warm-up and the timed region
    check0 = bes[0].host_frame() if (single and rank == 0 and not args.no_cpu_baseline and B == 1) else None
    for i in range(args.warmup):
        step(i)
    flush()
    torch.cuda.synchronize()
    for b in bes:
        b.device_synchronize()
        b.drain_timing()
        b.set_option("timing", 1 if (F == 1 and B == 1) else 0)  # per-kernel events only where kernels of different frames cannot overlap
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sync_ms[0] = 0.0
    frame0_time = frame_no[0]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        if i == 0 and check0 is not None:
            bes[0].download_frame(check0, accumulator=True)
    flush()  # a last, shorter batch when --steps is not a multiple of --batch: exactly K frames are timed
    if args.readback:
        for b in bes:
            b.wait_downloads()
    for b in bes:
        b.device_synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays_total = sum(view_rays(i) for i in range(args.steps))
    last_frame_time = frame_no[0] - 1

    # ---- what the timed region left behind
    timed_frames_gpu = None
    if check0 is not None:
        bes[0].wait_downloads()
        lb, lframes = state["last"]
        timed_frames_gpu = [(0, np.array(check0, copy=True)), (lframes[-1], lb.accumulator())]
    # Sharded frame: is what the all-gather + assemble left on this rank the frame one GPU renders alone?  (static scenes only)
    shard_check = None
    if world > 1 and rank == 0 and not animated and state["last"] is not None:
        lb, lframes = state["last"]
        whole = HipBackend.init(w, h, 1.0, device=dev, max_path_length=args.max_path_length)
        scene.mark_all_changed()
        scene.sync(whole)
        shard_check = True
        pres = [whole.host_frame(presented=True), lb.host_frame(presented=True)] if gather_format == 2 else None
        for f, vi in enumerate(lframes):
            whole.reset_accumulation()
            whole.render(views[vi % N_VIEWS])
            if gather_format == 0:    # the accumulators themselves
                same = np.array_equal(lb.accumulator_at(f).view(np.uint32), whole.accumulator().view(np.uint32))
            elif gather_format == 1:  # the finished frame in halves: what one GPU's finished frame rounds to
                same = np.array_equal(lb.framebuffer_at(f)[..., :3].astype(np.float16).view(np.uint16), whole.framebuffer()[..., :3].astype(np.float16).view(np.uint16))
            else:                     # the presented frame, byte for byte
                whole.download_frame(pres[0]); whole.wait_downloads()
                lb.download_frame(pres[1], frame=f); lb.wait_downloads()
                same = np.array_equal(pres[0], pres[1])
            shard_check = shard_check and bool(same)
        whole.close()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # ---- N > 1: the same frames in batches of 8 — ONE exchange per batch instead of one per frame — as a secondary mode (north_star's protocol,
    # one exchange per frame, is the `value` above)
    batch_mode = None
    if world > 1 and B == 1 and not args.no_modes and not animated and not args.emulate_shard and not args.readback:
        for b_ in bes:
            b_.close()
        headline_setup = (B, F, use_slots)
        B, F = 8, 3
        use_slots = native and os.environ.get("RFW_BENCH_INSTANCES") is None
        inst = make_instances(F, B, use_slots)
        connect(inst)
        bes = [i[0] for i in inst]
        state.update({"issued": 0, "pending": [], "k": 0, "last": None})
        nfb = max(args.mode_frames - args.mode_frames % B, 2 * B)
        for i in range(2 * B):
            step(i)
        flush()
        for b_ in bes:
            b_.device_synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        t_b = time.perf_counter()
        for i in range(nfb):
            step(i)
        flush()
        for b_ in bes:
            b_.device_synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        el_b = time.perf_counter() - t_b
        t = torch.tensor([el_b], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_b = float(t[0])
        batch_mode = {"Mrays_per_s": round(sum(view_rays(i) for i in range(nfb)) / el_b / 1e6, 1), "ms_per_frame": round(el_b / nfb * 1e3, 4), "frames": nfb,
                      "exchanges": nfb // B, "note": "extension call rfw_hip_render_batch: 8 frames traced per launch, ONE exchange of their 8 framebuffers"}
        B, F, use_slots = headline_setup  # (what the headline ran with: the line below describes THAT)
    # host cost of the per-frame scene update (animate + set_3d_instances + synchronize) without back-pressure
    host_sync_ms = None
    host_render_ms = None
    if animated:
        acc_t = acc_r = 0.0
        for i in range(20):
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            scene.animate(frame_no[0] / 60.0)
            frame_no[0] += 1
            scene.sync(bes[0])
            t_r = time.perf_counter()
            acc_t += t_r - t_s
            bes[0].render(views[0])   # (the frame slot's own instance update — staging copy, one upload, two launches — and the frame's four launches)
            acc_r += time.perf_counter() - t_r
        bes[0].device_synchronize()
        host_sync_ms = acc_t / 20 * 1e3
        host_render_ms = acc_r / 20 * 1e3

    ms_step = elapsed / args.steps * 1e3
    value = rays_total / elapsed / 1e6
    headline_mode = ("render() per frame" if B == 1 else f"render_batch of {B}") + (f", {F} frame slots" if use_slots else (f", {F} instances" if F > 1 else ", one frame at a time"))

    # ---- secondary modes on one GPU (after the timed region; same views, fresh instances so that buffer sizes match the mode)
    modes = {}
    iso = None
    if rank == 0:
        modes[headline_mode] = {"Mrays_per_s": round(value, 1), "ms_per_frame": round(ms_step, 4), "frames": args.steps, "is_value": True}
        if world > 1 and B == 1:
            modes[headline_mode]["exchanges"] = args.steps  # one per frame
        if batch_mode:
            modes["render_batch of 8, 3 batches in flight: one exchange per batch (extension call)"] = batch_mode
    if single and not args.no_modes and not animated:
        if F == 1 and B == 1:
            iso = bes[0].drain_timing()  # the timed region itself was one frame at a time: its own per-kernel events
        for b_, _, _ in inst:
            b_.close()
        inst, bes = [], []

        def run_mode(slots, batch, frames, timing, spp=1):
            ins = make_instances(max(slots, 1), max(batch, spp), slots > 1)
            mb = ins[0][0]
            mb.set_option("timing", 1 if timing else 0)
            seq = list(range(frames))

            def go(ids):
                if spp > 1:  # `spp` samples of ONE image per call (C4's "4 spp"): one launch per stage over all samples
                    for i in ids:
                        mb.render_samples(views[i % N_VIEWS], spp)
                elif batch > 1:
                    for j in range(0, len(ids), batch):
                        mb.render_batch([views[i % N_VIEWS] for i in ids[j:j + batch]])
                else:
                    for i in ids:
                        if slots <= 1 or args.identical_frames:
                            mb.reset_accumulation()
                        mb.render(views[i % N_VIEWS])
            go(seq[:max(batch * max(slots, 1), 8)])  # warm
            mb.device_synchronize()
            mb.drain_timing()
            t_s = time.perf_counter()
            go(seq)
            mb.device_synchronize()
            el = time.perf_counter() - t_s
            ms, n = mb.drain_timing() if timing else ({}, 0)
            mb.close()
            rays = sum(view_rays(i) for i in seq) * spp  # (samples 1 .. spp - 1 trace within a fraction of a percent of sample 0's ray count)
            return {"Mrays_per_s": round(rays / el / 1e6, 1), "ms_per_frame": round(el / frames * 1e3, 4), "frames": frames}, ms, n

        nf_mode = max(args.mode_frames, 16)
        if not (F == 1 and B == 1):
            m, iso_ms, iso_n = run_mode(1, 1, nf_mode, True)
            modes["render() per frame, one frame at a time"] = m
            iso = (iso_ms, iso_n)
        if not (B == 1 and use_slots and F == 8):
            modes["render() per frame, 8 frame slots"] = run_mode(8, 1, nf_mode, False)[0]
        if not (B == 8 and use_slots and F == 3):
            modes["render_batch of 8, 3 frame slots (extension call)"] = run_mode(3, 8, nf_mode - nf_mode % 8, False)[0]
        if args.max_path_length > 1:
            modes["render_samples: 4 spp of one image per call, 3 frame slots (extension call)"] = run_mode(3, 1, max(nf_mode // 4, 8), False, spp=4)[0]
    elif F == 1 and B == 1 and single:
        ms, n = bes[0].drain_timing()
        iso = (ms, n)
    elif single:
        # no separate strict mode requested: a short one-frame-at-a-time pass on the instance at hand gives the per-kernel durations
        # (C3: with its scene update before every frame, as in the timed region)
        bes[0].set_option("timing", 1)
        bes[0].drain_timing()
        for i in range(24):
            if animated:
                animate_and_sync(bes[0])
            bes[0].reset_accumulation()
            bes[0].render(views[i % N_VIEWS])
            bes[0].device_synchronize()
        iso = bes[0].drain_timing()

    # this job's own HBM roofline: a float4 device copy, measured after the timed region
    bw_measured = None
    if rank == 0:
        try:
            pb = bes[0] if bes else HipBackend.init(64, 64, 1.0, device=dev)
            bw_measured = pb.bandwidth_probe(1 << 30, 20)
            # ... and its vector-issue ceilings: v_fma_f32 alone, and the instruction mix of the node test (rfw_hip.h: rfw_hip_issue_probe)
            ISSUE_MEASURED.update({"fma_only": round(pb.issue_probe(0), 1), "per_lane_node_test_mix": round(pb.issue_probe(1), 1),
                                   "packet_node_step_mix": round(pb.issue_probe(2), 1)})
            if not bes:
                pb.close()
        except Exception:
            bw_measured = None

    if rank == 0:
        node_b, tri_b = sstats["node_bytes"], sstats["tri_bytes"]
        nv = len(per_view)
        mean = lambda key, k=None: sum((s[key][k] if k is not None else s[key]) for s in per_view) / nv
        n_prim, n_shad = mean("primary_rays"), mean("shadow_rays")
        # SURVEY §8(d)'s contract figure: ALGORITHMIC bytes per launch (mean over the views)
        alg = {
            # nodes x 64 B + triangles x 48 B + instance records x 64 B + 48 B written per ray (origin, direction, hit) + 16 B accumulator clear
            "k_primary": mean("nodes_visited", 0) * node_b + mean("tris_tested", 0) * tri_b + mean("instances_entered", 0) * 64 + n_prim * (48 + 16),
            # + 32 B queue entry read per ray, 16 B contribution read and 32 B accumulator read-modify-write per unoccluded ray (counted for all)
            "k_shadow": mean("nodes_visited", 2) * node_b + mean("tris_tested", 2) * tri_b + mean("instances_entered", 2) * 64 + n_shad * (32 + 16 + 32),
            # per path: hit 16 + ray 32 read; per hit: RTTriangle 176 + material 96 + normal matrix 48; shadow-queue push 48 per shadow ray
            "k_shade": n_prim * (16 + 32) + n_prim * (176 + 96 + 48) + n_shad * 48,
        }
        n_ext = mean("extension_rays")
        if n_ext:
            # the extension rays of ALL bounces of a frame: 32 B of ray read and 16 B of hit record written per ray
            alg["k_extend"] = mean("nodes_visited", 1) * node_b + mean("tris_tested", 1) * tri_b + mean("instances_entered", 1) * 64 + n_ext * (32 + 16)
        kms = None
        if iso and iso[1] > 0:
            kms = {"k_primary": iso[0]["ms_trace_primary"] / iso[1], "k_shadow": iso[0]["ms_trace_shadow"] / iso[1], "k_shade": iso[0]["ms_shade"] / iso[1]}
            if "k_extend" in alg:
                kms["k_extend"] = iso[0]["ms_trace_extend"] / iso[1]
            # launches of a kernel per frame: the ms above are sums over them, the counters of profiles/ are means per launch
            # (shadow rays: bounce 0 is traced one ray per lane by k_shadow, the bounces by the streaming flavour k_shadow_stream)
            launches = {"k_primary": 1, "k_shade": args.max_path_length, "k_shadow": 1, "k_shadow_stream": max(args.max_path_length - 1, 0),
                        "k_extend": max(args.max_path_length - 1, 1)}
        roofline = build_roofline(alg, kms, ms_step, sum(alg.values()), bw_measured, args, single, launches if kms else None)
        if os.environ.get("RFW_PACKET_TRACE", "1") not in ("0", "2"):
            # camera rays walk the tree as wavefront packets (csrc/traverse_packet.h): a node (128 B, the octant copy) and a triangle packet (48 B)
            # are fetched ONCE per wavefront through the scalar cache — what the launch actually asks the memory system for, next to §8(d)'s
            # per-ray figure above (which prices every ray's own visits at 64 B per node)
            waves = max(n_prim / 64.0, 1.0)
            roofline["primary_packets"] = {"node_steps_per_wavefront": round(mean("node_test_executions", 0) / waves, 2),
                                           "triangle_steps_per_wavefront": round(mean("tri_test_executions", 0) / waves, 2),
                                           "bytes_fetched_per_launch": int(mean("node_test_executions", 0) * 128 + mean("tri_test_executions", 0) * 48 + n_prim * (48 + 16))}
        roofline["nodes_per_ray"] = {"primary": round(mean("nodes_visited", 0) / max(n_prim, 1), 2), "shadow": round(mean("nodes_visited", 2) / max(n_shad, 1), 2)}
        roofline["tris_per_ray"] = {"primary": round(mean("tris_tested", 0) / max(n_prim, 1), 2), "shadow": round(mean("tris_tested", 2) / max(n_shad, 1), 2)}
        # SIMD efficiency of the traversal, from the instrumented frames: active lanes / 64 per execution of the node test and of the
        # triangle test, and what lanes that finished before their wavefront cost (nodes / (64 x max per wave))
        roofline["lane_utilisation"] = {name: {"node_test": round(mean("nodes_visited", k) / max(64 * mean("node_test_executions", k), 1), 3),
                                               "triangle_test": round(mean("tris_tested", k) / max(64 * mean("tri_test_executions", k), 1), 3),
                                               "finished_lanes_bound": round(mean("nodes_visited", k) / max(64 * mean("wave_max_nodes", k), 1), 3),
                                               # share of the wavefront-level node tests in which all active lanes visit ONE node
                                               "wave_uniform_node_tests": round(mean("uniform_node_test_executions", k) / max(mean("node_test_executions", k), 1), 3)}
                                        for k, name in ((0, "primary"), (1, "extension"), (2, "shadow")) if mean("node_test_executions", k)}
        out = {
            "metric": "Mrays/s (primary+shadow, 1spp)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": ranks_rendered,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic atrium ({'glTF scene' if not args.procedural else 'procedural'}), {sstats['triangles']} triangles in {sstats['instances']} instance(s), {w}x{h}, 1 spp, "
                                   + ("primary+shadow (max path length 1)" if args.max_path_length == 1 else f"path traced, max path length {args.max_path_length}, NEE")
                                   + (", every instance moved and the TLAS rebuilt on the device every frame" if animated else ", static scene") + ", BVH4",
                       "scene_source": scene_source, "mode": headline_mode,
                       "views": "one view (identical frames)" if args.identical_frames else ("one view, the scene changes every frame" if animated else f"{N_VIEWS} camera views cycled (5 mm dolly per frame): no two frames in flight are the same image"),
                       "rays_per_frame": int(sum(rays_of_view_total) / len(rays_of_view_total)), "rays_timed": int(rays_total),
                       "frames_in_flight": F * B, "batches_in_flight": F, "frames_per_batch": B,
                       "frames_in_flight_held_by": "frame slots of one instance (one scene copy)" if use_slots else (f"{F} renderer instances" if F > 1 else "-"),
                       "modes": modes,
                       "readback_every_frame": args.readback or False, "sharded_frame_equals_single_gpu_frame": shard_check,
                       "ranks_seen": ranks_seen, "ranks_rendered": ranks_rendered, "distinct_devices": distinct_devices, "ranks": ranks_report,
                       "launched_by": ("bench.py itself (--gpus N without a launcher: N child processes)" if os.environ.get("RFW_BENCH_LAUNCHED_BY") else
                                       ("an external launcher (WORLD_SIZE in the environment)" if world > 1 else "-")),
                       "dist_backend": (("nccl (RCCL)" if dist_backend == "nccl" else dist_backend + " (TEST HOOK: ranks may share a device)") if world > 1 else None),
                       "tile_shard": "64x64 round-robin" if world > 1 else "none", "collective": (args.collective if world > 1 else None),
                       "gather_format": (args.gather_format if (world > 1 or args.emulate_shard) else None), "present_rank": (args.present_rank if world > 1 else None),
                       "gather_bytes_per_frame": ({"f32": 12, "f16": 6, "bgra8": 4}[args.gather_format] * w * h if world > 1 else None),
                       "bvh": {"blas_nodes": sstats["blas_nodes"], "node_bytes": node_b, "tri_bytes": tri_b, "build_ms": round(sstats["ms_blas_build"], 1), "build": build_report,
                               # spatial splits (duplicates of the few triangles whose boxes waste the most) and what the acceleration structures take
                               # in device memory as allocated (nodes, their per-octant copies, the packet form of those where a packet kernel can run, 48-B packets)
                               "split_references": sstats.get("split_references"), "accel_bytes": sstats.get("accel_bytes"),
                               "accel_GB_per_million_triangles": round(sstats.get("accel_bytes", 0) / 1e9 / max(sstats["triangles"] / 1e6, 1e-9), 3),
                               "packet_copies": bool(sstats.get("packet_copies"))},
                       "synchronize_s": round(sync_s, 2), "max_path_length": args.max_path_length,
                       "instances": sstats["instances"], "tlas_nodes": sstats["tlas_nodes"],
                       "per_frame_synchronize_ms": round(host_sync_ms, 3) if animated else None,
                       "per_frame_render_call_ms": round(host_render_ms, 3) if animated else None,
                       "per_frame_synchronize_wall_ms_in_timed_region": round(sync_ms[0] / args.steps, 3) if animated else None},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: the contract's CPU leg)
            check = []
            if timed_frames_gpu is not None:
                check = [(views[vi % N_VIEWS], (frame0_time + k_ * (args.steps - 1)) if animated else None) for k_, (vi, _) in enumerate(timed_frames_gpu)]
            cb, frames_cpu = cpu_baseline(scene, check if check else [(views[0], None)], w, h, args.cpu_seconds, args.max_path_length, animated)
            out["cpu_baseline"] = cb
            if timed_frames_gpu is not None:
                names = ("first", "last")
                res = {}
                for name, (vi, g_), c_ in zip(names, timed_frames_gpu, frames_cpu):
                    same = bool(np.array_equal(g_.view(np.uint32), c_.view(np.uint32)))
                    rel = float(np.linalg.norm(g_.astype(np.float64) - c_.astype(np.float64)) / max(np.linalg.norm(c_.astype(np.float64)), 1e-30))
                    res[name] = {"frame": int(vi) if name == "first" else args.steps - 1, "bit_identical": same, "rel_l2": rel}
                res["all"] = all(v["bit_identical"] for v in res.values())
                out["config"]["timed_frame_equals_oracle"] = res
        out["config"]["modes"].update(baseline_rows)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for b in bes:
        b.close()


def baseline_configs(args):
    """BASELINE.json's other single-GPU configurations, each as a short run of this script in a child process (one after the other, before this
    process touches the device; nothing is exec'ed): C2 (Sponza-class, 262 k triangles, primary + shadow), C3 (C2 + 10 000 animated instances, the
    TLAS rebuilt every frame) and C4 as BASELINE.json words it (the 1 M-triangle scene path traced with NEE, max path length 3; 4 spp per call as
    one of its modes).  Every child checks its own first and last timed frame against the oracle and times the oracle on its own workload."""
    import subprocess
    rows = {}
    todo = (("BASELINE config 2 (Sponza-class 262 k triangles, primary+shadow, static BVH)", ["--workload", "atrium262k"], False),
            ("BASELINE config 3 (config 2 + 10 000 animated instances, TLAS rebuilt every frame)", ["--workload", "spheres10k"], False),
            ("BASELINE config 4 (~1 M triangles path traced with NEE, max path length 3)", ["--workload", "atrium1m", "--max-path-length", "3"], True))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if not QUEUES_SET_BY_CALLER:
        env.pop("GPU_MAX_HW_QUEUES", None)  # (this process chose it for ITS configuration: a child chooses for its own — C3 runs 20 slots over 24 queues)
    for name, extra, with_modes in todo:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.baseline_steps), "--warmup", "20", "--no-baseline-configs",
               "--cpu-seconds", "3", "--mode-frames", "32"] + extra + ([] if with_modes else ["--no-modes"]) + (["--procedural"] if args.procedural else [])
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                rows[name] = {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-400:]}
                continue
            d = json.loads(line[-1])
        except Exception as e:  # a child that fails costs its row, not the headline
            rows[name] = {"error": repr(e)[:300]}
            continue
        c = d["config"]
        row = {"Mrays_per_s": d["value"], "ms_per_frame": d["ms_per_step"], "frames": d["steps"], "workload": c["workload"], "mode": c["mode"],
               "rays_per_frame": c["rays_per_frame"], "timed_frame_equals_oracle": c.get("timed_frame_equals_oracle"),
               "cpu_baseline_Mrays_per_s": (d.get("cpu_baseline") or {}).get("value"), "cpu_baseline_cores": (d.get("cpu_baseline") or {}).get("cores"),
               "per_frame_synchronize_ms": c.get("per_frame_synchronize_ms"), "child_seconds": round(time.time() - t0, 1),
               "command": "python3 bench.py " + " ".join(cmd[2:])}
        if with_modes:
            row["modes"] = {k: v["Mrays_per_s"] for k, v in c.get("modes", {}).items() if not v.get("is_value")}
        rows[name] = row
    return rows


def latest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None


def profile_kernel(kernels, name):
    """The entry of kernel `name` in a committed PMC summary: the non-counting, single-frame instantiation(s) as the profiler prints them,
    counters SUMMED over the instantiations one launch of the stage consists of (the summaries hold means per launch):
      k_primary        the packet flavour when that is what ran (option packet_trace), else the one-ray-per-lane kernel
      k_shadow         bounce 0: the packet flavour = one launch per visiting order (<false, true> + <false, false>), else k_shadow<false, false>
      k_shadow_stream  the bounces' streaming launches: one per visiting order as well (both are counted)
      k_extend         the streaming flavour when that is what ran
      k_shade          either workgroup size (launch_shade picks by what shares the chip)"""
    def pick(*names):
        found = [kernels[n] for n in names if n in kernels]
        return found
    cands = {
        "k_primary": [("k_primary_packet<false>",), ("k_primary<false>",)],
        "k_shadow": [("k_shadow_packet<false, true>", "k_shadow_packet<false, false>"), ("k_shadow<false, false>",), ("k_shadow<false,false>",)],
        "k_shadow_stream": [("k_shadow_stream<false, true>", "k_shadow_stream<false, false>")],
        "k_extend": [("k_extend_stream<false>",), ("k_extend<false>",)],
        # (workgroups of 512 threads one frame at a time — what the counter passes run — and of 256 with frames in flight: the same instructions)
        "k_shade": [("k_shade<false, 512>",), ("k_shade<false, 256>",), ("k_shade<false>",)],
    }.get(name, [(name + "<false, false>",), (name + "<false,false>",), (name + "<false>",), (name,)])
    for group in cands:
        found = pick(*group)
        if not found:
            continue
        if len(found) == 1:
            return found[0]
        merged = {}
        for k in set().union(*found):
            vals = [f[k] for f in found if isinstance(f.get(k), (int, float))]
            if not vals:
                continue
            merged[k] = sum(vals) / len(vals) if "rate" in k else sum(vals)
        return merged
    return None


def build_roofline(alg, kms, ms_step, alg_frame, bw_measured, args, single, launches=None):
    """Which ceiling bounds the dominant kernel: every candidate as a fraction <= 1 of its own peak, the highest one is `bound`.

    Live in this run: the kernels' mean launch durations (HIP events on the launch stream, one frame at a time) and the algorithmic
    byte counts.  From the committed rocprofv3 PMC summaries of this workload (profiles/*_pmc_valu.json, *_pmc.json, *_pmc_cache.json —
    counters cannot be read from inside the process): wave64 VALU instructions, vector-memory read instructions, L2 requests and
    HBM-side bytes per launch.  Instruction counts are properties of (scene, view, kernel code), so dividing them by the live
    duration is legitimate as long as the profile is of the same code; the profile's tag is named in `counters_from`."""
    cfg = config_key(args) if single else None
    pv, pm, pc = (latest_profile(f"*_{cfg}_pmc_valu.json"), latest_profile(f"*_{cfg}_pmc.json"), latest_profile(f"*_{cfg}_pmc_cache.json")) if cfg else (None, None, None)
    usable = bool(cfg) and any((pv, pm, pc))
    r = {"unit": "GB/s", "counters_from": [p[1] for p in (pv, pm, pc) if p] if usable else None}
    if usable:
        # counters measured on other kernel sources say nothing about this build: refuse them rather than print ceilings of another program
        here = kernel_source_hash()
        stale = [p[1] for p in (pv, pm, pc) if p and p[0].get("source_hash") != here]
        if stale:
            usable = False
            r["counters_from"] = None
            r["stale_counters"] = {"files": stale, "note": "measured on other kernel sources than this checkout's (source_hash differs): not used; re-run tools/measure.sh"}
    launches = launches or {}
    # a path-traced frame with frames in flight runs the STREAMING flavour of k_extend / k_shadow for the bounces (one frame at a time: one ray
    # per lane): the timed region's ceilings come from the counter set measured with streaming forced (profiles/*_<cfg>S_*), the isolated
    # kernel's from the plain set
    sets = {"isolated": (pv, pm, pc), "timed": (pv, pm, pc)}
    if usable and cfg == "c4path":
        sv, sm, sc_ = latest_profile(f"*_{cfg}S_pmc_valu.json"), latest_profile(f"*_{cfg}S_pmc.json"), latest_profile(f"*_{cfg}S_pmc_cache.json")
        if all((sv, sm, sc_)) and all(p[0].get("source_hash") == here for p in (sv, sm, sc_)):
            sets["timed"] = (sv, sm, sc_)
            r["counters_from"] += [p[1] for p in (sv, sm, sc_)]
    if not kms or not any(kms.values()):
        r.update({"bound": None, "kernel": None, "achieved": None, "peak": None, "frac": None, "traffic": None,
                  "note": "no isolated per-kernel durations in this configuration (kernels of frames in flight overlap)"})
        return r
    dom = max(kms, key=lambda k: kms[k])
    traffic = None

    def ceilings_of(names, seconds, which="isolated"):
        """Every ceiling as a fraction <= 1 for the launches `names` (one launch each) taking `seconds` together."""
        nonlocal traffic
        c = {}
        if not usable:
            return c
        pv, pm, pc = sets[which]
        tot = {"valu": 0.0, "vmem": 0.0, "ta_busy": 0.0, "l2": 0.0, "l2_hit": [], "hbm": 0.0, "l1_hit": []}
        have = {"valu": True, "ta": True, "l2": True, "hbm": True}
        for n in names:
            kv = profile_kernel(pv[0]["kernels"], n) if pv else None
            km = profile_kernel(pm[0]["kernels"], n) if pm else None
            kc = profile_kernel(pc[0]["kernels"], n) if pc else None
            nl = launches.get(n, 1)  # launches of this kernel per frame (the summaries hold means per launch)
            if kv and kv.get("SQ_INSTS_VALU"):
                tot["valu"] += nl * kv["SQ_INSTS_VALU"]; tot["vmem"] += nl * kv.get("SQ_INSTS_VMEM_RD", 0)
            else:
                have["valu"] = False
            if kc and kc.get("TA_BUSY_avr"):
                tot["ta_busy"] += nl * kc["TA_BUSY_avr"]
                if kc.get("l1_hit_rate") is not None:
                    tot["l1_hit"].append(kc["l1_hit_rate"])
            else:
                have["ta"] = False
            if kc and kc.get("l2_request_bytes_per_launch"):
                tot["l2"] += nl * kc["l2_request_bytes_per_launch"]; tot["l2_hit"].append(kc.get("l2_hit_rate"))
            else:
                have["l2"] = False
            if km:
                tot["hbm"] += nl * km["hbm_bytes_per_launch_corrected"]
            else:
                have["hbm"] = False
        if have["valu"] and tot["valu"]:
            a = tot["valu"] / seconds / 1e9
            c["valu_issue"] = {"achieved": round(a, 1), "peak": VALU_PEAK_GIPS, "unit": "G wave64 VALU instructions/s", "frac": round(a / VALU_PEAK_GIPS, 4),
                               "per_launch": int(tot["valu"])}
            if ISSUE_MEASURED.get("fma_only"):
                # the guide's peak is one v_fma_f32 per 2 cycles per SIMD at 2.4 GHz.  Three rates measured live in this job (rfw_hip_issue_probe,
                # 8 wavefronts per SIMD, chip-wide): v_fma_f32 alone; the instruction mix of ONE CHILD OF THE PER-LANE NODE TEST (k_shadow,
                # k_extend, k_primary without packets: byte conversions, packed FMAs, min / max, compares); the mix of ONE NODE STEP OF THE
                # PACKET KERNEL (k_primary_packet: FMAs with a scalar operand, min / max, one compare per child, its scalar instructions beside
                # them).  The kernels are made of MORE than their node test (triangle tests, shading: mostly full-rate multiplies and adds), so
                # a kernel's own rate can lie above the rate of its node test's mix: the ratios below compare, they are not fractions of a ceiling.
                c["valu_issue"]["measured_rates"] = dict(ISSUE_MEASURED, unit="G wave64 VECTOR instructions/s, chip-wide, 8 wavefronts per SIMD",
                                                         ratio_to_per_lane_node_test_mix=round(a / ISSUE_MEASURED["per_lane_node_test_mix"], 4),
                                                         ratio_to_packet_node_step_mix=round(a / ISSUE_MEASURED["packet_node_step_mix"], 4),
                                                         frac_of_fma_only=round(a / ISSUE_MEASURED["fma_only"], 4))
        if have["ta"] and tot["ta_busy"]:
            # the texture-address / vector-L1 path of a CU, the unit every 16-B-per-lane load goes through: busy cycles (mean over the CUs)
            # of the launches over the cycles they took.  Its cost per wave instruction grows with the cache lines the 64 lanes touch
            # (tools/probes/mem_probe.hip: ~11 cycles when all lanes read one line, ~75 when every lane reads its own)
            busy_s = tot["ta_busy"] / (CLOCK_GHZ * 1e9)
            c["l1_ta"] = {"achieved": round(busy_s * 1e3, 4), "peak": round(seconds * 1e3, 4), "unit": "ms busy (TA_BUSY, mean over CUs) of ms elapsed", "frac": round(busy_s / seconds, 4),
                          "vmem_read_instructions": int(tot["vmem"]), "l1_hit_rate": (round(sum(tot["l1_hit"]) / len(tot["l1_hit"]), 4) if tot["l1_hit"] else None)}
        if have["l2"] and tot["l2"]:
            a = tot["l2"] / seconds / 1e9
            c["l2"] = {"achieved": round(a, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(a / L2_PEAK_GBS, 4), "hit_rate": tot["l2_hit"], "per_launch_bytes": int(tot["l2"])}
        if have["hbm"] and tot["hbm"]:
            a = tot["hbm"] / seconds / 1e9
            c["hbm"] = {"achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4), "per_launch_bytes": int(tot["hbm"]),
                        "note": "L2 fabric-side requests (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE): Infinity-Cache hits are included, so true HBM bytes are lower still"}
            if len(names) == 1:
                traffic = int(tot["hbm"] / max(launches.get(names[0], 1), 1))  # per launch, like `achieved`
        return c

    dur = kms[dom] * 1e-3
    ceilings = ceilings_of([dom], dur)
    # the timed region as a whole: one launch of every kernel of a frame over the wall time per frame (frames in flight overlap, so this is
    # the utilisation the chip actually runs at, where the per-kernel figures above are launches with the machine to themselves)
    steady = ceilings_of(["k_primary", "k_shade", "k_shadow", "k_assemble"] + (["k_extend", "k_shadow_stream"] if "k_extend" in kms else []), ms_step * 1e-3, "timed")
    nl_dom = max(launches.get(dom, 1), 1)
    contract = {"algorithmic_bytes_per_launch": int(alg[dom] / nl_dom), "avg_launch_ms": round(kms[dom] / nl_dom, 4), "launches_per_frame": nl_dom,
                "algorithmic_GBps": round(alg[dom] / dur / 1e9, 1), "hbm_peak_GBps": HBM_PEAK_GBS,
                "measured_copy_GBps": round(bw_measured, 1) if bw_measured else None,
                "note": "SURVEY §8(d)'s figure: bytes the rays of one launch ask for, lane by lane.  64 coherent lanes share most node fetches, so this is an UPPER bound on traffic, "
                        "not traffic: it is reported as a rate next to the peak, not as a fraction of it",
                "reuse_factor_vs_measured_traffic": round(alg[dom] / traffic, 1) if traffic else None,
                "per_kernel": {k: {"ms": round(kms[k], 4), "algorithmic_GBps": round(alg[k] / (kms[k] * 1e-3) / 1e9, 1) if kms[k] > 0 else None} for k in alg},
                "timed_region_algorithmic_GBps": round(alg_frame / (ms_step * 1e-3) / 1e9, 1)}
    if ceilings:
        # The line's `frac` is ALWAYS the same quantity (it used to be whichever unit read highest): HBM-side bytes of the dominant
        # kernel by the rocprofv3 counters (FETCH_SIZE x 2 + WRITE_SIZE, per launch) / that kernel's launch duration measured live / the 8 TB/s peak.
        # Which unit actually binds the kernel — it is instruction issue, not memory — is `binding_unit`, with every ceiling under `ceilings`.
        binding = max(ceilings, key=lambda k: ceilings[k]["frac"])
        hbm = ceilings.get("hbm")
        r.update({"bound": "hbm", "kernel": dom, "achieved": hbm["achieved"] if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": hbm["frac"] if hbm else None, "traffic": traffic,
                  "frac_is": "HBM-side bytes per launch of the dominant kernel (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, profiles/) / its live launch duration / 8 TB/s",
                  "binding_unit": {"name": binding, "frac": ceilings[binding]["frac"], "achieved": ceilings[binding]["achieved"], "peak": ceilings[binding]["peak"],
                                   "unit": ceilings[binding]["unit"],
                                   "note": "the ceiling this kernel sits closest to (valu_issue: wave64 vector instructions per second; l1_ta: busy cycles of the texture-address unit); "
                                           "the kernels are bound by instruction issue, which is why the HBM fraction is small: the scene lives in L2 + Infinity Cache"},
                  "ceilings": ceilings, "contract": contract})
        if steady:
            sb = max(steady, key=lambda k: steady[k]["frac"])
            r["timed_region"] = {"hbm_frac": steady["hbm"]["frac"] if "hbm" in steady else None, "binding_unit": sb, "binding_frac": steady[sb]["frac"],
                                 "ms_per_frame": round(ms_step, 4), "ceilings": steady}
        if ISSUE_MEASURED.get("fma_only") and "valu_issue" in ceilings:
            # the guide's vector peak against what THIS device issued in THIS job: the one rate that IS a ceiling for any mix is v_fma_f32 alone
            r["measured_peak"] = {"fma_only": ISSUE_MEASURED["fma_only"], "unit": ceilings["valu_issue"]["unit"]}
            r["frac_of_measured_peak"] = {"kernel_alone": round(ceilings["valu_issue"]["achieved"] / ISSUE_MEASURED["fma_only"], 4)}
            if steady and "valu_issue" in steady:
                r["frac_of_measured_peak"]["timed_region"] = round(steady["valu_issue"]["achieved"] / ISSUE_MEASURED["fma_only"], 4)
    else:
        # no committed counters for this configuration: only the contract's rate can be given; `frac` stays null rather than a number above 1
        r.update({"bound": None, "kernel": dom, "achieved": contract["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "frac": None, "traffic": None, "contract": contract})
    return r


def usable_cpus():
    """Threads this process may actually run at once: the scheduler affinity mask capped by the cgroup CPU quota (cpu.max) — os.cpu_count()
    reports the machine, and a pool sized from it on a quota-limited box only adds context switches."""
    limits = {"os_cpu_count": os.cpu_count() or 1}
    try:
        limits["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        limits["sched_affinity"] = limits["os_cpu_count"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    limits["cgroup_cpu_quota"] = round(quota, 2) if quota else None
    n = limits["sched_affinity"]
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, limits


def cpu_baseline(scene, check, w, h, budget_s, max_path_length, animated):
    """The oracle — a CPU RESTATEMENT of the reference's rtbvh / MBVH path, not rfw-rs measured — timed on this host's cores on the
    bench's own workload at the bench's own resolution: the frames of `check` (view, animation time) first, whose accumulators are
    returned for the comparison with the timed frames, then more frames of the same views until about `budget_s` seconds are spent."""
    from oracle.bindings import Oracle, build_timing_library
    cores, limits = usable_cpus()
    # the same source, built for speed on THIS machine (-O3 -march=native; -ffp-contract=off stays: its frames are still the checker's frames)
    timing_lib, timing_flags = build_timing_library()
    orc = Oracle(w, h, library=timing_lib, threads=cores, max_path_length=max_path_length)
    scene.mark_all_changed()
    t0 = time.time()
    scene.sync(orc)
    build_s = time.time() - t0
    frames = []
    n, spent, rays, busy = 0, 0.0, 0, 0
    count = lambda s_: s_["primary"] + s_["shadow"] + s_["extension"]
    k = 0
    while True:
        view, t_anim = check[k % len(check)]
        if animated and t_anim is not None and k < len(check):
            scene.animate(t_anim / 60.0)
            scene.mark_all_changed()
            scene.sync(orc)
        orc.reset()
        before = count(orc.stats())   # (reset may or may not clear the ray counters: count this frame's rays as a difference)
        t1 = time.perf_counter()
        orc.render(view)
        spent += time.perf_counter() - t1
        rays += count(orc.stats()) - before
        busy = max(busy, orc.stats().get("busy_threads", 0))
        n += 1
        if k < len(check):
            frames.append(orc.accumulator().copy())
        k += 1
        if k >= len(check) and (spent > budget_s or n >= 64):
            break
    # thread scaling of the same frame (view 0), one frame per point: says whether the full pool is limited by the cores or by something
    # else (quota, NUMA, memory) — per-thread rate at `all` within 2x of the single-thread rate = the pool scales
    scaling = []
    rate_all = rays / spent / 1e6
    for nt in sorted({1, min(8, cores), min(64, cores)}):
        if nt >= cores:
            continue
        orc.set_option("threads", nt)
        orc.set_option("tile_stride", max(1, 16 // nt))  # a bounded sample: every 16th tile of the frame for one thread, every 2nd for eight
        orc.reset()
        before = count(orc.stats())
        t1 = time.perf_counter()
        orc.render(check[0][0])
        dt = time.perf_counter() - t1
        r_ = (count(orc.stats()) - before) / dt / 1e6
        scaling.append({"threads": nt, "Mrays_per_s": round(r_, 3), "krays_per_s_per_thread": round(1e3 * r_ / nt, 1), "seconds": round(dt, 2)})
    orc.set_option("threads", cores)
    orc.set_option("tile_stride", 1)
    scaling.append({"threads": busy or cores, "Mrays_per_s": round(rate_all, 3), "krays_per_s_per_thread": round(1e3 * rate_all / (busy or cores), 1)})
    single = scaling[0]["krays_per_s_per_thread"] if scaling and scaling[0]["threads"] == 1 else None
    per_thread = scaling[-1]["krays_per_s_per_thread"]
    verdict = None
    if single:
        ratio = single / max(per_thread, 1e-9)
        verdict = (f"per-thread rate with all threads is {ratio:.2f}x below the single-thread rate: " +
                   ("the pool scales with the cores" if ratio <= 2.0 else
                    ("limited by the cgroup CPU quota" if limits.get("cgroup_cpu_quota") and limits["cgroup_cpu_quota"] < limits["sched_affinity"] else
                     "limited by something other than core count (SMT siblings, NUMA / memory bandwidth: BVH and triangles are shared by all threads)")))
    return ({"value": round(rate_all, 3), "unit": "Mrays/s", "cores": busy or cores, "kind": "port",
             "what": "CPU restatement of the reference's path (oracle/), not the reference binary: rfw-rs cannot be built here (no Rust toolchain, rtbvh un-vendored)",
             "build": (f"g++ {timing_flags}, built on this host" if timing_lib else f"oracle/liboracle.so (-O2, portable): {timing_flags}"),
             "threads": {"started": cores, "rendered_at_least_one_tile": busy, "work_items": "16x16-pixel tiles from one atomic counter, persistent pool; per-thread counters on cache lines of their own",
                         "limits": limits, "scaling": scaling, "scaling_verdict": verdict},
             "sample": f"{n} frame(s) of the bench's own scene and views at {w}x{h}, 1 spp, max path length {max_path_length}, {busy or cores} busy threads of {cores}, {spent:.1f} s of rendering; BVH build {build_s:.1f} s excluded",
             },
            frames)


if __name__ == "__main__":
    main()
