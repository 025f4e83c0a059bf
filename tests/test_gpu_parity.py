"""Parity proper: the HIP path through the C ABI against the oracle on the same seeded inputs.
Ray queries (ids, t, u, v) must be bit-exact; accumulated radiance must match to <= 1e-4 relative L2
(north_star tolerance) — the arithmetic contract actually yields identical bits, which is asserted too."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative L2 per image, BASELINE.json north_star


def make(kind, w, h, a=0, b=0, c=0.0, seed=1, **opts):
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build(kind, a, b, c, seed)
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, **{k: v for k, v in opts.items() if k in ("max_path_length", "flags", "builder", "streams", "frames_in_flight", "max_batch")})
    scene.sync(be)
    orc = Oracle(w, h, threads=8)
    if "max_path_length" in opts:
        orc.set_option("max_path_length", opts["max_path_length"])
    scene.mark_all_changed()
    scene.sync(orc)
    return scene, be, orc


def random_rays(n, seed, extent=4.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def assert_hits_equal(g, r):
    assert np.array_equal(g["inst"], r["inst"]), np.flatnonzero(g["inst"] != r["inst"])[:10]
    assert np.array_equal(g["tri"], r["tri"]), np.flatnonzero(g["tri"] != r["tri"])[:10]
    hit = r["inst"] >= 0
    for f in ("t", "u", "v"):
        assert np.array_equal(g[f][hit].view(np.uint32), r[f][hit].view(np.uint32)), f


@pytest.mark.parametrize("kind,a,b", [("cornell", 0, 0), ("soup", 3000, 1), ("soup", 800, 12)])
def test_closest_hit_bit_exact(kind, a, b):
    scene, be, orc = make(kind, 64, 64, a, b, seed=7)
    o, d = random_rays(20000, 11)
    # add the camera's own primary rays (coherent) and axis-parallel rays (inf / nan slab arithmetic)
    po, pd = orc.primary_rays(scene.view(64, 64), 0)
    ax = np.zeros((6, 3), np.float32)
    for i in range(6):
        ax[i, i % 3] = 1.0 if i < 3 else -1.0
    o = np.concatenate([o, po, np.zeros((6, 3), np.float32)])
    d = np.concatenate([d, pd, ax])
    g = be.intersect(o, d)
    assert_hits_equal(g, orc.intersect(o, d))                # oracle through its MBVH
    assert_hits_equal(g, orc.intersect(o, d, brute=True))    # and the tree-free definition


@pytest.mark.parametrize("kind,a,b", [("cornell", 0, 0), ("soup", 3000, 1), ("soup", 800, 12)])
def test_any_hit_exact(kind, a, b):
    scene, be, orc = make(kind, 32, 32, a, b, seed=5)
    o, d = random_rays(20000, 13)
    tmax = np.random.default_rng(2).uniform(0.05, 8.0, size=len(o)).astype(np.float32)
    g = be.occludes(o, d, tmax)
    assert np.array_equal(g, orc.occludes(o, d, tmax))
    assert np.array_equal(g, orc.occludes(o, d, tmax, brute=True))
    # metamorphic: occluded <=> the closest hit (same t_min) lies before t_max
    h = be.intersect(o, d, t_min=1e-3)
    assert np.array_equal(g.astype(bool), (h["inst"] >= 0) & (h["t"] < tmax))


def test_edge_cases_empty_and_removed_instances():
    from rfw_rs_amd import HipBackend, Scene, pod
    import ctypes as C
    be = HipBackend.init(16, 16)
    be.synchronize()  # nothing uploaded at all
    o, d = random_rays(100, 1)
    assert (be.intersect(o, d)["inst"] == -1).all()
    assert (be.occludes(o, d, np.full(100, 10.0, np.float32)) == 0).all()
    be.render(Scene().build("cornell").view(16, 16))  # render before any mesh: a no-op, as gpu-rt (lib.rs:1686-1688)
    # zero matrix = removed instance slot (crates/rfw-scene/src/instances_3d.rs:79-86)
    scene, be2, orc = make("soup", 16, 16, 200, 3, seed=3)
    md = scene.mesh_data(0)
    mats = (pod.Mat4 * 3)()
    for i in range(3):
        for k in (0, 5, 10, 15):
            mats[i].m[k] = 1.0
        mats[i].m[12] = 2.5 * i
    for k in range(16):
        mats[1].m[k] = 0.0
    inst = pod.InstancesData3D()
    inst.local_aabb = md.bounds
    inst.matrices = mats
    inst.num_matrices = 3
    for b in (be2, ):
        b.set_3d_instances(0, inst)
        b.synchronize()
    orc._l.orc_set_3d_instances(orc._h, 0, C.byref(inst))
    orc._l.orc_synchronize(orc._h)
    o, d = random_rays(5000, 4, extent=6.0)
    g, r = be2.intersect(o, d), orc.intersect(o, d, brute=True)
    assert_hits_equal(g, r)
    assert set(np.unique(g["inst"])) <= {-1, 0, 2}


@pytest.mark.parametrize("kind,a,b,mpl,spp", [("cornell", 0, 0, 1, 1), ("cornell", 0, 0, 3, 4), ("soup", 1500, 6, 3, 2), ("gallery", 0, 0, 3, 3)])
def test_radiance_matches_oracle(kind, a, b, mpl, spp):
    w, h = 96, 64
    scene, be, orc = make(kind, w, h, a, b, seed=9, max_path_length=mpl)
    view = scene.view(w, h)
    for _ in range(spp):
        be.render(view)
        orc.render(view)
    ga, ra = be.accumulator(), orc.accumulator()
    assert np.isfinite(ga).all()
    assert rel_l2(ga, ra) <= TOL
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), f"{(ga != ra).sum()} differing floats"
    assert rel_l2(be.framebuffer(), orc.framebuffer()) <= TOL
    s, os_ = be.frame_stats(), orc.stats()
    assert s["sample_count"] == spp
    # ray counts of the last frame: the oracle counts all frames
    assert s["primary_rays"] == w * h


def test_counts_and_queues_match_oracle_one_frame():
    w, h = 64, 64
    scene, be, orc = make("soup", w, h, 1200, 4, seed=2, max_path_length=3)
    view = scene.view(w, h)
    be.render(view)
    orc.render(view)
    s, o = be.frame_stats(), orc.stats()
    assert (s["primary_rays"], s["extension_rays"], s["shadow_rays"]) == (o["primary"], o["extension"], o["shadow"])


def test_queue_counters_are_those_of_the_latest_frame():
    """Round 6: the queue counters live in a ring of two blocks per instance — a frame's k_primary clears the block of the NEXT frame instead
    of a memset in front of every frame.  Whatever the number of frames rendered before, the counts read back are the latest frame's: two
    views alternate (their ray counts differ), several samples accumulate, and a counting frame sits in between."""
    from rfw_rs_amd import pod
    w, h = 64, 64
    scene, be, orc = make("soup", w, h, 1200, 4, seed=2, max_path_length=3)
    va = scene.view(w, h)
    vb = pod.CameraView3D.from_buffer_copy(va)
    vb.pos.x += 0.4; vb.p1.x += 0.4
    want = {}
    for name, v in (("a", va), ("b", vb)):
        orc.reset(); orc.render(v)
        o = orc.stats()
        want[name] = (o["primary"], o["extension"], o["shadow"])
    assert want["a"] != want["b"]
    got = lambda: tuple(be.frame_stats()[k] for k in ("primary_rays", "extension_rays", "shadow_rays"))
    for k, name in enumerate("abbabaab"):
        be.reset_accumulation()
        be.render(va if name == "a" else vb)
        assert got() == want[name], (k, name)
        if k == 3:
            be.set_option("count_traversal", 1); be.reset_accumulation(); be.render(vb)
            assert got() == want["b"] and sum(be.frame_stats()["nodes_visited"]) > 0
            be.set_option("count_traversal", 0)
    be.close()


def test_accumulation_reset_on_camera_change():
    w, h = 32, 32
    scene, be, orc = make("cornell", w, h)
    v1 = scene.view(w, h)
    be.render(v1); be.render(v1)
    assert be.frame_stats()["sample_count"] == 2
    scene.set_camera([0.1, 0, -3.4], [0, 0, 1], 40.0, 0.0, 1.0)
    be.render(scene.view(w, h))
    assert be.frame_stats()["sample_count"] == 1
    be.reset_accumulation()
    be.render(scene.view(w, h))
    assert be.frame_stats()["sample_count"] == 1


@pytest.mark.parametrize("tile_size", [32, 24, 128])   # Z-ordered 4 x 4 blocks, row-major 3 x 3 blocks (not a power of two), Z-ordered 16 x 16
def test_tile_sharding_is_bit_exact(tile_size):
    """1 GPU == N shards: each rank renders its tiles into a slab; gathered + assembled frame == the unsharded frame."""
    import torch
    from rfw_rs_amd import HipBackend, Scene
    w, h = 200, 136  # not a multiple of the tile size: ragged edge tiles
    scene = Scene().build("soup", 1500, 5, 0.0, 4)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    full = HipBackend.init(w, h, 1.0)
    scene.sync(full)
    for _ in range(2):
        full.render(view)
    ref = full.accumulator()
    world = 3
    ranks = []
    for r in range(world):
        be = HipBackend.init(w, h, 1.0, rank=r, world=world, tile_size=tile_size)
        scene.mark_all_changed()
        scene.sync(be)
        ranks.append(be)
    slab = ranks[0].shard_info()["slab_floats"]
    gathered = torch.zeros(world, slab, dtype=torch.float32, device="cuda")
    for r, be in enumerate(ranks):
        be.set_slab_output(gathered[r].data_ptr())
        for _ in range(2):
            be.render(view)
        be.device_synchronize()
    ranks[0].assemble_frame(gathered.data_ptr())
    got = ranks[0].accumulator()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the numpy twin of the kernels' slab indexing (rfw-rs_amd/dist.py, used by the CPU tests of the N > 1 path) names the same slots
    from rfw_rs_amd import dist as rd
    for r in range(world):
        want = rd.extract_slab(ref[..., :3], r, world, tile_size)
        assert np.array_equal(gathered[r].cpu().numpy().reshape(-1, 3).view(np.uint32), want.view(np.uint32)), r


@pytest.mark.parametrize("fmt", [1, 2])
def test_finished_frame_gather_formats_equal_the_single_gpu_frame(fmt):
    """VERDICT r02 #6a/b: only the FINISHED frame has to travel — halves (option gather_format = 1, 6 B per pixel) or the presented B, G, R, A
    bytes (2, 4 B per pixel) instead of the accumulator's floats (12 B) — and only the rank that presents has to de-tile it at once
    (present_rank).  Three shards on one device: what every rank ends up with equals the single-GPU frame in the same format, bit for bit."""
    import torch
    from rfw_rs_amd import BackendError, HipBackend, Scene
    w, h = 200, 136
    scene = Scene().build("soup", 1500, 5, 0.0, 4)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    full = HipBackend.init(w, h, 1.0)
    scene.sync(full)
    for _ in range(2):
        full.render(view)
    ref_frame = full.framebuffer()
    ref_presented = full.host_frame(presented=True)
    full.download_frame(ref_presented)
    full.wait_downloads()
    world = 3
    ranks = []
    for r in range(world):
        be = HipBackend.init(w, h, 1.0, rank=r, world=world, tile_size=32)
        be.set_option("gather_format", fmt)
        be.set_option("present_rank", 0)
        scene.mark_all_changed()
        scene.sync(be)
        ranks.append(be)
    words = ranks[0].shard_info()["slab_floats"]
    px_per_slab = ranks[0].shard_info()["tiles_local"] * 32 * 32
    assert words == (px_per_slab * 3 // 2 if fmt == 1 else px_per_slab)          # 6 / 4 bytes per pixel instead of 12
    gathered = torch.zeros(world, words, dtype=torch.float32, device="cuda")     # opaque 4-byte words
    for r, be in enumerate(ranks):
        be.set_slab_output(gathered[r].data_ptr())
        for _ in range(2):
            be.render(view)
        be.device_synchronize()
    for be in ranks:                                                              # rank 0 de-tiles now, ranks 1 and 2 when somebody reads
        be.assemble_frame(gathered.data_ptr())
    for r, be in enumerate(ranks):
        if fmt == 1:
            got = be.framebuffer()[..., :3]
            want = ref_frame[..., :3].astype(np.float16).astype(np.float32)       # the device's conversion rounds to nearest even, as numpy's
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), r
        else:
            dst = be.host_frame(presented=True)
            be.download_frame(dst)
            be.wait_downloads()
            assert np.array_equal(dst, ref_presented), r
            with pytest.raises(BackendError):
                be.framebuffer()                                                  # only the presented frame exists
        with pytest.raises(BackendError):
            be.accumulator()                                                      # the accumulators stayed on their ranks
    for be in ranks + [full]:
        be.close()


def test_full_size_properties():
    """BASELINE config C2 at full size (1920x1080, ~262k triangles): properties that need no oracle run."""
    from rfw_rs_amd import HipBackend, Scene
    w, h = 1920, 1080
    scene = Scene().build("atrium", 262267, 0, 0.0, 0xC0FFEE)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=1)
    scene.sync(be)
    be.render(view)
    a1 = be.accumulator().copy()
    s = be.frame_stats()
    assert s["primary_rays"] == w * h and 0 < s["shadow_rays"] <= w * h
    be.reset_accumulation()
    be.render(view)
    assert np.array_equal(a1.view(np.uint32), be.accumulator().view(np.uint32))  # idempotent / deterministic
    assert np.isfinite(a1).all() and a1[..., :3].max() <= 10.0 * 2 + 1e-3       # clamp 10 per contribution, <= 2 adds per pixel
    # closest-hit vs any-hit consistency on the frame's own primary rays
    from oracle.bindings import Oracle
    po, pd = Oracle(w // 8, h // 8).primary_rays(scene.view(w // 8, h // 8), 0)
    hit = be.intersect(po, pd, t_min=1e-3)
    tm = np.where(hit["inst"] >= 0, hit["t"], np.float32(50.0)).astype(np.float32)
    assert (be.occludes(po, pd, tm * np.float32(1.01) + np.float32(1e-3)) == (hit["inst"] >= 0)).all()
    assert (be.occludes(po, pd, tm * np.float32(0.99)) == 0).sum() >= 0.999 * len(po)


@pytest.mark.parametrize("gather_format", ["f32", "f16", "bgra8"])
def test_bench_two_ranks_on_one_gpu_gloo_hook(gather_format):
    """bench.py's N > 1 code path (tile shard, all-gather, assemble, max-over-ranks timing) with two processes sharing this GPU
    through the gloo test hook (RCCL itself refuses two ranks on one device), for every format the tiles can travel in."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RFW_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "cornell", "--width", "320", "--height", "200", "--no-cpu-baseline", "--gather-format", gather_format]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["rays_per_frame"] >= 320 * 200
    # the value is north_star's protocol — one exchange per FRAME; batches of 8 (one exchange per batch) are a row of config.modes
    assert out["config"]["frames_per_batch"] == 1 and out["config"]["sharded_frame_equals_single_gpu_frame"] is True
    rows = out["config"]["modes"]
    assert any(v.get("is_value") and v.get("exchanges") == 3 for v in rows.values()), rows
    assert any("render_batch of 8" in k and v["exchanges"] * 8 == v["frames"] and v["Mrays_per_s"] > 0 for k, v in rows.items()), rows
    assert out["config"]["gather_format"] == gather_format
    assert out["config"]["gather_bytes_per_frame"] == {"f32": 12, "f16": 6, "bgra8": 4}[gather_format] * 320 * 200
    assert "distinct" in out["config"]["views"] or "cycled" in out["config"]["views"]


def test_plain_bench_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher around it and no WORLD_SIZE in the environment must run two ranks (the
    driver's scaling command may be exactly that): bench.py starts them as child processes before it touches the GPU, and the line says who
    rendered.  Two ranks share this box's one device through the gloo hook."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RFW_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cornell", "--width", "320", "--height", "200",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line: rank 0's
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["ranks_rendered"] == 2 and c["sharded_frame_equals_single_gpu_frame"] is True
    assert sorted(x["rank"] for x in c["ranks"]) == [0, 1] and len({x["pid"] for x in c["ranks"]}) == 2
    assert all(x["rays_of_view_0"] > 0 and x["tiles"] > 0 for x in c["ranks"])
    assert "bench.py itself" in c["launched_by"] and c["collective"] == "torch" and "gloo" in c["dist_backend"]
    # ... and one rank stays one rank whatever a stray WORLD_SIZE says
    env1 = dict(env, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "cornell", "--width", "320",
                        "--height", "200", "--no-cpu-baseline", "--no-modes"], env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["ranks_seen"] == 1 and out["config"]["ranks"][0]["rank"] == 0


@pytest.mark.parametrize("present_rank", [0, -1])
def test_bench_two_ranks_on_one_gpu_p2p(present_rank):
    """bench.py --collective p2p with one process per rank (both on this GPU): the handles travel through torch.distributed once, the
    peers' buffers are mapped with hipIpcOpenMemHandle, the tiles travel as stores and the ranks meet in each other's flag words only.
    With every rank a destination (-1) both ranks de-tile every frame."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RFW_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--workload", "cornell", "--width", "320", "--height", "200", "--no-cpu-baseline", "--collective", "p2p", "--present-rank", str(present_rank)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["config"]["collective"] == "p2p" and out["config"]["sharded_frame_equals_single_gpu_frame"] is True
    assert "frame slots" in out["config"]["frames_in_flight_held_by"]


def test_scale_script_dry_run_covers_every_exchange_and_format():
    """tools/scale.sh --dry: the rehearsal of the 1 / 2 / 4 / 8 GPU table on one device — two ranks per (exchange, format) pair, torch's
    collective and the peer-store exchange (IPC handles; also with fine-grained flag words and with the cached receive buffer), every
    sharded frame compared with the single-GPU frame.  The first multi-GPU run must not fail for boring reasons."""
    import os
    import subprocess
    from conftest import ROOT
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale.sh"), "--dry"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert r.stdout.count("ok   ") == 10, r.stdout[-2000:]


_CONTENDED_BUILDS = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
from rfw_rs_amd import HipBackend, Scene
w, h = 256, 144
scene = Scene().build("atrium", 1048576, int(sys.argv[4]), 0.0, 0xC0FFEE)   # 0: two meshes (built one after the other); 1: 65 meshes (one forest)
scene.set_aspect(w / h)
view = scene.view(w, h)
for k in range(int(sys.argv[2])):
    be = HipBackend.init(w, h, 1.0, builder=int(sys.argv[3]))
    scene.mark_all_changed(); scene.sync(be)
    be.render(view)
    print(hashlib.sha1(be.accumulator().tobytes()).hexdigest(), flush=True)
    be.close()
"""


def test_device_builder_while_another_process_uses_the_device():
    """Regression (round 3): the level kernels of the binned-SAH builder run their wavefronts in any order, and one of them cleared the
    NEXT level's bins in the array another was still reading THIS level's bins from — invisible while a launch's wavefronts start together,
    a memory fault or a hang (one build in five) as soon as a second process time-slices the device, which is how two ranks share a GPU.
    Two processes build the 1 M-triangle scene six times each, side by side — as two meshes and as 65 (one forest) —; every build must give
    the image of the host builder's tree."""
    import subprocess
    import sys
    from conftest import ROOT
    for meshes in ("0", "1"):       # the scene as two meshes, and as 65 (the forest build)
        ref = subprocess.run([sys.executable, "-c", _CONTENDED_BUILDS, ROOT, "1", "1", meshes], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert ref.returncode == 0, ref.stderr[-2000:]
        want = ref.stdout.split()[-1]
        procs = [subprocess.Popen([sys.executable, "-c", _CONTENDED_BUILDS, ROOT, "6", "3", meshes], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]
            assert out.split() == [want] * 6, meshes


def test_animated_instances_match_oracle():
    """C3 in miniature: a grid of icosphere instances moved every frame -> set_3d_instances + synchronize + render."""
    w, h = 96, 64
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("cornell").build("spheres", 6, 5, 0.3)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    orc = Oracle(w, h, threads=4, max_path_length=2)
    for frame in range(3):
        scene.animate(frame / 3.0)
        scene.sync(be)
        scene.mark_all_changed()
        scene.sync(orc)
        orc.reset()
        be.render(view)
        orc.render(view)
        assert be.frame_stats()["sample_count"] == 1   # the scene changed: accumulation restarted
        ga, ra = be.accumulator(), orc.accumulator()
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), frame
    assert be.scene_stats()["instances"] == 31


@pytest.mark.parametrize("builder", [1, 2, 3])  # DEVICE_SAH = 3 (BLAS by binned SAH on the device); HOST_SAH (both levels on the host), DEVICE_LBVH (both levels on the device); 0 = AUTO is the default elsewhere
def test_builders_give_identical_answers(builder):
    """Metamorphic: the image and the ray queries are functions of the scene, not of the acceleration structure."""
    w, h = 96, 64
    scene, be, orc = make("soup", w, h, 1500, 9, seed=21, max_path_length=3, builder=builder)
    o, d = random_rays(20000, 31)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    tmax = np.random.default_rng(7).uniform(0.05, 8.0, size=len(o)).astype(np.float32)
    assert np.array_equal(be.occludes(o, d, tmax), orc.occludes(o, d, tmax, brute=True))
    view = scene.view(w, h)
    for _ in range(2):
        be.render(view)
        orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))


@pytest.mark.parametrize("builder", [1, 2, 3])
@pytest.mark.parametrize("tau", [0.0, 8e-5, 1e-7])
def test_spatial_splits_never_change_the_image(builder, tau):
    """Round 5: the few triangles whose boxes waste the most (walls of two triangles across the atrium) are referenced several times, each
    reference with the tight box of a part of the triangle; duplicates report the id of the triangle they stand for, so hits, ties and images
    are those of the unsplit scene — whatever the threshold (off; the default; one far below it: the outlier rule decides), for every
    builder, after a full build, after an edit of one mesh (the incremental path) and with the 65 meshes of C4 built as one forest."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 160, 104
    cases = [(40000, 0), (40000, 1)] + ([(262267, 0)] if (builder == 3 and tau == 8e-5) else [])  # (the C2 scene: hundreds of duplicates)
    for n_target, sphere_meshes in cases:
        scene = Scene().build("atrium", n_target, sphere_meshes, 0.0, 0xC0FFEE)
        scene.set_aspect(w / h)
        view = scene.view(w, h)
        be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder)
        be.set_option("spatial_splits", tau)
        scene.sync(be)
        orc = Oracle(w, h, threads=8, max_path_length=3)
        scene.mark_all_changed(); scene.sync(orc)
        st = be.scene_stats()
        assert st["triangles"] == orc.stats()["n_tris"]           # the caller's triangles: duplicates are not counted
        assert (st["split_references"] > 0) == (tau > 0.0), st
        if n_target > 100000:
            assert 300 <= st["split_references"] <= 1024, st
        o, d = random_rays(20000, 5, extent=12.0)
        assert_hits_equal(be.intersect(o, d), orc.intersect(o, d))
        tmax = np.random.default_rng(7).uniform(0.05, 30.0, size=len(o)).astype(np.float32)
        assert np.array_equal(be.occludes(o, d, tmax), orc.occludes(o, d, tmax))
        for _ in range(2):
            be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
        # everything re-sent (registered host arrays, heads first), then rendered again
        scene.mark_all_changed(); scene.sync(be)
        be.reset_accumulation(); orc.reset()
        be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
        assert be.scene_stats()["split_references"] == st["split_references"]
        be.close()


@pytest.mark.parametrize("group", [0, 256, 512])
@pytest.mark.parametrize("slots", [1, 3])
def test_shade_workgroup_size_never_changes_the_image(group, slots):
    """Round 5: k_shade runs in workgroups of 256 threads where frames overlap (several frame slots, one frame per call) and of 512 otherwise
    (option "shade_group": 0 = that rule).  The queues a workgroup files its rays into are compacted per workgroup, so the ORDER of the shadow
    and extension queues depends on the size — the image must not: path traced frames, single and as a batch, against the oracle."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 200, 136  # (ragged against the 256 / 512-path groups and the 128-pixel tiles)
    scene = Scene().build("atrium", 30000, 1, 0.0, 0xBEEF)
    scene.set_aspect(w / h)
    views = []
    for origin in ([0.5, 2.0, 9.0], [-3.0, 1.5, 6.0], [4.0, 3.0, 7.0]):
        scene.look_at(origin, [0.0, 1.0, 0.0])
        views.append(scene.view(w, h))
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=slots, max_batch=3)
    be.set_option("shade_group", group)
    scene.sync(be)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    scene.mark_all_changed(); scene.sync(orc)
    for v in views[:2]:
        be.reset_accumulation(); orc.reset()
        for _ in range(2):
            be.render(v); orc.render(v)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    if slots == 1:
        be.render_batch(views)
        for f, v in enumerate(views):
            orc.reset(); orc.render(v)
            assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    with pytest.raises(Exception):
        be.set_option("shade_group", 128)
    be.close()


@pytest.mark.parametrize("forest", [True, False])
def test_many_meshes_are_built_in_one_pass(forest, monkeypatch):
    """A scene of many meshes of very different sizes (2-triangle walls, boxes, an icosphere mesh with 30 instances, a 30 000-triangle soup,
    the 65 meshes of C4's atrium in miniature): the full device build takes all meshes as roots of ONE level-by-level pass
    (sah_build_forest) — or, with RFW_NO_FOREST, mesh by mesh on parallel lanes.  Ray queries equal the tree-free definition, the image
    equals the oracle's, and one mesh edited afterwards (the per-mesh path, in place) still does."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    if not forest:
        monkeypatch.setenv("RFW_NO_FOREST", "1")
    w, h = 160, 104
    scene = Scene().build("cornell").build("spheres", 6, 5, 0.3).build("soup", 30000, 3, 0.0, 5)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3)
    scene.sync(be)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    scene.mark_all_changed(); scene.sync(orc)
    o, d = random_rays(8000, 5)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    for _ in range(2):
        be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    st = be.scene_stats()
    assert st["triangles"] == orc.stats()["n_tris"] and st["blas_nodes"] > 0
    # everything changed again (the full build once more), then render: same image
    scene.mark_all_changed(); scene.sync(be)
    be.reset_accumulation(); orc.reset()
    be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.close()


def test_device_builders_on_large_mesh_match_host_sah():
    from rfw_rs_amd import HipBackend, Scene
    w, h = 480, 270
    scene = Scene().build("atrium", 262267, 0, 0.0, 0xC0FFEE)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    accs = []
    for builder in (1, 2, 3):
        be = HipBackend.init(w, h, 1.0, max_path_length=2, builder=builder)
        scene.mark_all_changed()
        scene.sync(be)
        be.render(view)
        accs.append(be.accumulator().copy())
        st = be.scene_stats()
        assert st["triangles"] > 250000 and st["blas_nodes"] > 0
        be.close()
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))   # host SAH == device LBVH
    assert np.array_equal(accs[0].view(np.uint32), accs[2].view(np.uint32))   # host SAH == device SAH


@pytest.mark.parametrize("streams", [1, 3, 8])
def test_substreams_do_not_change_the_image(streams):
    """A frame split into sub-shards on separate HIP streams is the same frame."""
    w, h = 200, 136
    scene, be, orc = make("soup", w, h, 1200, 5, seed=17, max_path_length=3, streams=streams)
    view = scene.view(w, h)
    for _ in range(2):
        be.render(view)
        orc.render(view)
    assert be.frame_stats()["substreams"] == streams
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    s, o = be.frame_stats(), orc.stats()
    assert s["primary_rays"] == w * h


@pytest.mark.parametrize("builder", [0, 1, 2])
def test_skinned_meshes_match_oracle_across_poses(builder):
    """set_skins -> skinned copies of the mesh per (mesh, skin) pair on the device -> BLAS rebuilt every
    synchronize.  The deformed triangles, the ray queries against them and the image are bit-identical to the oracle's."""
    w, h = 128, 96
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("skinned", 0, 0, 0.0, 3)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, builder=builder)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    o, d = random_rays(20000, 5, extent=3.0)
    po, pd = orc.primary_rays(view, 0)
    o, d = np.concatenate([o, po]), np.concatenate([d, pd])
    for frame, t in enumerate((0.7, 0.0, 1.9)):
        scene.pose(t)
        scene.sync(be)
        scene.mark_all_changed()
        scene.sync(orc)
        rt = orc.triangles()
        gt = be.debug_read("triangles", rt.nbytes).view(np.float32).reshape(-1, 44)
        assert be.scene_stats()["triangles"] == len(rt) == 1350
        assert np.array_equal(gt.view(np.uint32), rt.view(np.uint32)), (frame, np.argwhere(gt.view(np.uint32) != rt.view(np.uint32))[:5])
        g = be.intersect(o, d)
        assert_hits_equal(g, orc.intersect(o, d))
        assert_hits_equal(g, orc.intersect(o, d, brute=True))
        assert (g["tri"] >= 6 + 448).sum() > 100        # rays do land on the skinned copies
        orc.reset()
        be.render(view)
        orc.render(view)
        assert be.frame_stats()["sample_count"] == 1   # new pose: accumulation restarted
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), frame
    # a second sample of the same pose accumulates
    be.render(view)
    orc.render(view)
    assert be.frame_stats()["sample_count"] == 2
    ga, ra = be.accumulator(), orc.accumulator()
    assert rel_l2(ga, ra) <= TOL
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32))


def test_skin_set_change_relayouts_the_scene():
    """An instance that gains / loses its skin moves between the static mesh and a skinned copy."""
    w, h = 96, 64
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("cornell")
    be = HipBackend.init(w, h, 1.0, max_path_length=2)
    orc = Oracle(w, h, threads=4, max_path_length=2)
    view = scene.view(w, h)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    n0 = be.scene_stats()["triangles"]
    scene.build("skinned", 0, 0, 0.0, 3)   # adds the skinned tubes to the same scene
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    orc.reset()
    be.render(view); orc.render(view)
    assert be.scene_stats()["triangles"] > n0
    assert be.scene_stats()["triangles"] == orc.stats()["n_tris"]
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))


@pytest.mark.parametrize("which", ["room", "skinned", "textured"])
def test_gltf_scenes_match_oracle(tmp_path, which):
    """a glTF document through the C++ host's importer, then the same boundary calls into both backends."""
    from gltf_util import write_gltf, write_skinned_gltf, write_textured_gltf
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 96, 64
    path = {"room": lambda: write_gltf(tmp_path, "glb"), "skinned": lambda: write_skinned_gltf(tmp_path)[0],
            "textured": lambda: write_textured_gltf(tmp_path, True)[0]}[which]()
    scene = Scene().load_gltf(str(path))
    if which == "skinned":
        scene.build("cornell")                                  # something to light the strip
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3)
    orc = Oracle(w, h, threads=4, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o, d = random_rays(5000, 3, extent=3.0)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    for _ in range(2):
        be.render(view); orc.render(view)
    ga, ra = be.accumulator(), orc.accumulator()
    assert ra[..., :3].max() > 0 and orc.stats()["shadow"] > 0
    assert rel_l2(ga, ra) <= TOL
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32))


def test_obj_scene_matches_oracle(tmp_path):
    """A Wavefront OBJ with its material library and textures (PNG base colour, run-length TGA normal map, an emitter) through the host's
    ObjLoader restatement, lit by its own emissive pentagon and a Cornell box around it: same hits, same image."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    from test_obj import write_scene
    path, _, _ = write_scene(tmp_path)
    scene = Scene().load(str(path))
    scene.build("cornell")
    w, h = 96, 72
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3)
    orc = Oracle(w, h, threads=4, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o, d = random_rays(5000, 4, extent=3.0)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    for _ in range(2):
        be.render(view); orc.render(view)
    assert orc.stats()["shadow"] > 0
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.close()


@pytest.mark.parametrize("frames_in_flight", [0, 3])
def test_animated_gltf_matches_oracle_over_time(tmp_path, frames_in_flight):
    """examples/animated's set_animation_timers system (main.rs:221-223): Scene::set_animations_time every frame, then synchronize and
    render.  The document's animation moves a skinned tube (skinning + refit on the device) and a rigid cube (TLAS); its floor carries a
    JPEG base colour.  Every frame: triangles' hits and the accumulated image bit-identical to the oracle's."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    from gltf_util import write_animated_gltf
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    yy, xx = np.mgrid[0:64, 0:64]
    tex = np.stack([128 + 100 * np.sin(xx / 5.0), 128 + 90 * np.cos(yy / 7.0), xx * 4], -1).clip(0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(tex).save(b, "JPEG", quality=92)
    scene = Scene().load_gltf(str(write_animated_gltf(tmp_path, jpeg=b.getvalue())))
    w, h = 120, 80
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=frames_in_flight)
    orc = Oracle(w, h, threads=8, max_path_length=3)
    o, d = random_rays(4000, 9, extent=2.5)
    o[:, 1] = np.abs(o[:, 1]) * 0.6 + 0.1
    seen = []
    for t in (0.0, 0.31, 0.6, 1.0, 1.3, 1.77, 2.4):
        scene.set_animation_time(t)
        scene.sync(be)                                             # only what moved: changed instance lists, the skin
        scene.mark_all_changed()                                   # the second backend gets everything again
        scene.sync(orc)
        assert_hits_equal(be.intersect(o, d), orc.intersect(o, d))
        be.reset_accumulation(); orc.reset()
        for _ in range(2):
            be.render(view); orc.render(view)
        ga, ra = be.accumulator(), orc.accumulator()
        assert orc.stats()["shadow"] > 0 and ra[..., :3].max() > 0
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), t
        seen.append(ga.copy())
    assert not np.array_equal(seen[0], seen[2]) and not np.array_equal(seen[2], seen[4])
    be.close()


@pytest.mark.parametrize("builder", [1, 2, 3])
def test_coincident_primitives_do_not_break_the_builders(tmp_path, builder):
    """3000 copies of one triangle: every centroid in one bin on every axis, so the SAH builders fall back to halving ranges; the
    closest hit is the lowest triangle id among the exact ties."""
    from gltf_util import write_duplicates_gltf
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().load_gltf(str(write_duplicates_gltf(tmp_path)))
    be = HipBackend.init(32, 32, 1.0, builder=builder)
    orc = Oracle(32, 32, threads=2)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    assert be.scene_stats()["triangles"] == 3001
    rng = np.random.default_rng(2)
    o = np.stack([rng.uniform(-0.5, 3.5, 4000), rng.uniform(-0.5, 1.5, 4000), np.full(4000, -2.0)], axis=1).astype(np.float32)
    d = np.tile(np.array([0, 0, 1], np.float32), (4000, 1))
    g = be.intersect(o, d)
    assert_hits_equal(g, orc.intersect(o, d, brute=True))
    hit_dups = (g["tri"] >= 0) & (g["tri"] < 3000)
    assert hit_dups.sum() > 100 and np.all(g["tri"][hit_dups] == 0)      # exact ties -> lowest id
    assert (g["tri"] == 3000).sum() > 100


def test_frame_slots_pipeline_new_images_and_keep_accumulation_exact():
    """options.frames_in_flight: one instance, one scene, N frame slots.  New views go to the next slot; repeated views accumulate on
    their slot; every frame read back is bit-identical to the oracle's, whichever slot rendered it."""
    w, h = 96, 64
    scene, be, orc = make("soup", w, h, 1200, 4, seed=23, max_path_length=3, frames_in_flight=3)
    views = []
    for k in range(5):
        scene.set_camera([0.3 * k - 0.6, 0.4, -4.0 + 0.2 * k], [0.0, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    # five different views back to back: five new images over three slots, none waits for another
    for v in views:
        be.render(v)
    orc.reset(); orc.render(views[-1])
    assert be.frame_stats()["sample_count"] == 1
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    # the same view again: samples 2 and 3 accumulate on the slot that holds sample 1
    for _ in range(2):
        be.render(views[-1]); orc.render(views[-1])
    assert be.frame_stats()["sample_count"] == 3
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    fb = be.framebuffer()
    assert np.array_equal(fb.view(np.uint32), orc.framebuffer().view(np.uint32))
    # an earlier view again: a new image (its old slot may have been reused), one sample
    be.render(views[1]); orc.reset(); orc.render(views[1])
    assert be.frame_stats()["sample_count"] == 1
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    # reset_accumulation and a scene change both start new images
    be.render(views[1]); be.reset_accumulation(); be.render(views[1])
    assert be.frame_stats()["sample_count"] == 1
    scene.build("spheres", 3, 3, 0.5)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    for _ in range(2):
        be.render(views[2])
    orc.reset(); orc.render(views[2]); orc.render(views[2])
    assert be.frame_stats()["sample_count"] == 2
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    o, d = random_rays(2000, 4)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    be.close()


def test_frame_slots_with_instances_that_move_every_frame():
    """C3 in miniature through ONE instance with frame slots: set_3d_instances + synchronize + render every frame, each frame on the next
    slot with its own TLAS built from the lists of that moment; every frame must equal the oracle's for the same pose."""
    w, h = 96, 64
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("cornell").build("spheres", 6, 5, 0.3)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2, frames_in_flight=4)
    orc = Oracle(w, h, threads=4, max_path_length=2)
    wanted = []
    for frame in range(9):                      # nine frames queued without ever reading back
        scene.animate(frame / 3.0)
        scene.sync(be)
        be.render(view)
    scene.mark_all_changed(); scene.sync(orc); orc.reset(); orc.render(view)
    assert be.frame_stats()["sample_count"] == 1
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    for frame in (2, 5):                        # and frame by frame
        scene.animate(frame / 3.0)
        scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
        orc.reset(); be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), frame
        o, d = random_rays(3000, frame)
        assert_hits_equal(be.intersect(o, d), orc.intersect(o, d, brute=True))
    assert be.scene_stats()["instances"] == 31
    be.close()


@pytest.mark.parametrize("slots", [20, 24])
def test_many_frame_slots_with_instances_that_move_every_frame(slots):
    """BASELINE config 3 runs on 20 frame slots since round 6 (the library takes up to 24): 3 x slots frames queued without reading back,
    every one with its own pose and its own TLAS on its slot (the one-workgroup build: the instance has frame slots) — the last frame and
    three more, frame by frame, equal the oracle's.  (More slots than hardware queues only serialise: the runtime's default here is 4.)"""
    w, h = 80, 48
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("cornell").build("spheres", 9, 7, 0.3)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=2, frames_in_flight=slots)
    orc = Oracle(w, h, threads=4, max_path_length=2)
    n = 3 * slots
    for frame in range(n):
        scene.animate(frame / 5.0)
        scene.sync(be)
        be.render(view)
    scene.mark_all_changed(); scene.sync(orc); orc.reset(); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    for frame in (1, slots - 1, slots + 3):
        scene.animate(frame / 5.0)
        scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
        orc.reset(); be.render(view); orc.render(view)
        assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32)), frame
    assert int(be.debug_read("build_counters", 20).view(np.uint32)[4]) >= n     # every frame's TLAS went through the fused path
    be.close()


def test_frame_slots_with_skinned_meshes_and_resize():
    """Skinned copies live in the owner's shared mesh buffers: with them the slots share one TLAS and synchronize() waits for the frames in
    flight before it re-skins.  Every pose must still match the oracle; resize keeps working with slots."""
    w, h = 96, 64
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    scene = Scene().build("skinned", 0, 0, 0.0, 3)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=3)
    orc = Oracle(w, h, threads=4, max_path_length=3)
    for frame, t in enumerate((0.7, 0.2, 1.1, 1.9, 0.0)):   # pose, synchronize, render back to back: frames queue behind each other
        scene.pose(t)
        scene.sync(be)
        be.render(view)
    scene.mark_all_changed(); scene.sync(orc); orc.reset(); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.render(view); orc.render(view)                          # a second sample of the last pose
    assert be.frame_stats()["sample_count"] == 2
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    # resize: every slot gets new buffers, accumulation restarts
    w2, h2 = 64, 48
    be.resize((w2, h2))
    scene.set_aspect(w2 / h2)
    v2 = scene.view(w2, h2)
    orc2 = Oracle(w2, h2, threads=4, max_path_length=3)
    scene.mark_all_changed(); scene.sync(orc2)
    for _ in range(2):
        be.render(v2); orc2.render(v2)
    assert be.frame_stats()["sample_count"] == 2
    assert be.accumulator().shape == (h2, w2, 4)
    assert np.array_equal(be.accumulator().view(np.uint32), orc2.accumulator().view(np.uint32))
    be.close()


def batch_views(scene, w, h, k):
    views = []
    for i in range(k):
        scene.set_camera([0.35 * i - 0.7, 0.3 + 0.1 * i, -4.0 + 0.15 * i], [0.05 * i, 0.0, 1.0], fov=45.0, aspect=w / h)
        views.append(scene.view(w, h))
    return views


@pytest.mark.parametrize("kind,a,b,mpl,k", [("soup", 1500, 4, 3, 4), ("cornell", 0, 0, 2, 7), ("gallery", 0, 0, 4, 16)])
def test_render_batch_equals_single_renders(kind, a, b, mpl, k):
    """rfw_hip_render_batch: k independent new images in one launch per stage.  Frame f must be exactly what render(view f) on a reset
    instance produces — and so exactly the oracle's image of that view."""
    w, h = 104, 72  # ragged edge tiles
    scene, be, orc = make(kind, w, h, a, b, seed=31, max_path_length=mpl, max_batch=16)
    views = batch_views(scene, w, h, k)
    be.render_batch(views)
    assert be.frame_stats()["sample_count"] == 1
    for f, v in enumerate(views):
        orc.reset(); orc.render(v)
        assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
        assert np.array_equal(be.framebuffer_at(f).view(np.uint32), orc.framebuffer().view(np.uint32)), f
    # a plain render afterwards starts a new image (and accumulates from there), even for a view of the batch
    be.render(views[0]); be.render(views[0])
    orc.reset(); orc.render(views[0]); orc.render(views[0])
    assert be.frame_stats()["sample_count"] == 2
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    # a shorter batch after a longer one; more frames than max_batch is an error, not a truncation
    be.render_batch(views[1:3])
    orc.reset(); orc.render(views[2])
    assert np.array_equal(be.accumulator_at(1).view(np.uint32), orc.accumulator().view(np.uint32))
    be.render_batch(views[3:4]); be.render_batch(views[3:4])   # a batch of one is still a new image each time
    orc.reset(); orc.render(views[3])
    assert be.frame_stats()["sample_count"] == 1 and np.array_equal(be.accumulator_at(0).view(np.uint32), orc.accumulator().view(np.uint32))
    from rfw_rs_amd import BackendError
    with pytest.raises(BackendError):
        be.render_batch((views * 17)[:17])
    with pytest.raises(BackendError):
        be.accumulator_at(16)
    be.close()


def test_render_batch_through_frame_slots_and_shards():
    """A batch per frame slot (consecutive batches overlap), and a batch on a tile-sharded frame: the slab is [frame][slab], the gathered
    buffer [rank][frame][slab], one assemble for all frames."""
    import torch
    from rfw_rs_amd import HipBackend
    w, h = 120, 88
    scene, be, orc = make("soup", w, h, 900, 6, seed=12, max_path_length=3, max_batch=4, frames_in_flight=3)
    views = batch_views(scene, w, h, 8)
    be.render_batch(views[:4])
    be.render_batch(views[4:])
    for f in range(4):
        orc.reset(); orc.render(views[4 + f])
        assert np.array_equal(be.accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    be.close()
    world, k = 3, 3
    ranks = []
    for r in range(world):
        b = HipBackend.init(w, h, 1.0, rank=r, world=world, tile_size=32, max_path_length=3, max_batch=4)
        scene.mark_all_changed()
        scene.sync(b)
        ranks.append(b)
    slab = ranks[0].shard_info()["slab_floats"]
    gathered = torch.zeros(world, 4, slab, dtype=torch.float32, device="cuda")[:, :k].contiguous()
    for r, b in enumerate(ranks):
        b.set_slab_output(gathered[r].data_ptr())
        b.render_batch(views[:k])
        b.device_synchronize()
    ranks[1].assemble_batch(gathered.data_ptr(), k)
    for f in range(k):
        orc.reset(); orc.render(views[f])
        assert np.array_equal(ranks[1].accumulator_at(f).view(np.uint32), orc.accumulator().view(np.uint32)), f
    for b in ranks:
        b.close()


@pytest.mark.parametrize("size,array", [((12, 5), 1), ((100, 37), 1), ((100, 37), 0), ((1024, 1024), 1)])
def test_texture_array_normalisation_matches_oracle(tmp_path, size, array):
    """gpu-rt/src/lib.rs:1230-1246: material textures become 1024 x 1024 x 5-mip array layers (non-square, non-power-of-two PNGs
    included); with option texture_array = 0 they are sampled at their native size.  Either way the device equals the oracle."""
    from gltf_util import encode_png, write_textured_gltf
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    tw, th = size
    tex = np.random.default_rng(tw * 7 + th).integers(20, 255, size=(th, tw, 4), dtype=np.uint8)
    tex[..., 3] = 255
    path, _ = write_textured_gltf(tmp_path, True, png=encode_png(tex))
    scene = Scene().load_gltf(str(path))
    w, h = 128, 96
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3)
    orc = Oracle(w, h, threads=4, max_path_length=3)
    be.set_option("texture_array", array); orc.set_option("texture_array", array)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    for _ in range(2):
        be.render(view); orc.render(view)
    ga, ra = be.accumulator(), orc.accumulator()
    assert ra[..., :3].max() > 0 and ra[60:, :, :3].std() > 1e-3      # the textured floor is in view
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32))
    be.close()


@pytest.mark.parametrize("fmt,present_rank", [(0, -1), (1, 0), (2, 0), (2, -1)])
def test_library_exchange_through_the_loopback_hub(fmt, present_rank):
    """the exchange INSIDE the library (render() packs, gathers, de-tiles: what `bench.py --collective native` runs over RCCL
    on a multi-GPU node) cannot be rehearsed with world > 1 on one device through RCCL itself.  rfw_hip_comm_init_loopback stands in for the
    communicator only (tests/loopback_ranks.py: three ranks in one process, three frame slots each, six frames in flight, every gather format,
    a rank that never arrives).  In a process of its own: like several NCCL ranks driven by one process, every stream involved needs a
    hardware queue of its own (a rank's wait must not sit in front of the copies it waits for), and the runtime reads GPU_MAX_HW_QUEUES once."""
    import subprocess, sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="24")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback_ranks.py"), str(fmt), str(present_rank)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "LOOPBACK OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
