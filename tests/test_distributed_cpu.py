"""N > 1 path on CPU: two processes (gloo), tile-sharded slabs, ONE all-gather, assemble == the unsharded frame.
The renderer is stubbed by slicing a known frame (no GPU here); the GPU test test_tile_sharding_is_bit_exact checks the
kernels' tile dealing against the same numpy map."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RFW_ROOT"])
import numpy as np, torch, torch.distributed as dist
from rfw_rs_amd import dist as rd
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['RFW_PORT']}", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
w, h, ts, streams = 200, 136, 32, 4
rng = np.random.default_rng(42)
frame = rng.random((h, w, 4), dtype=np.float32)           # the frame every rank would agree on
slab = torch.from_numpy(rd.extract_slab(frame, rank, world, ts, streams))  # what THIS rank renders (4 sub-shards back to back)
gathered = rd.all_gather_slabs(slab)                       # the one collective per frame
out = rd.assemble(gathered.numpy(), w, h, ts, streams)
assert gathered.shape[0] == world and np.array_equal(out, frame), "assembled frame differs"
owner, slot = rd.slab_index_map(w, h, world, ts, streams)
assert set(np.unique(owner)) == set(range(world))
# every (owner, slot) pair addresses a distinct slab element: no two pixels collide
assert len(np.unique(owner * gathered.shape[1] + slot)) == w * h
# a batch of frames (rfw_hip_render_batch on a sharded frame): slab = [frame][tile pixels], ONE all-gather for the batch,
# gathered = [rank][frame][tile pixels]; frame f is assembled from gathered[:, f]
k = 3
frames = rng.random((k, h, w, 4), dtype=np.float32)
bslab = torch.from_numpy(np.stack([rd.extract_slab(frames[f], rank, world, ts, 1) for f in range(k)]))
bg = rd.all_gather_slabs(bslab)  # (world, k, slab_elems, 4)
for f in range(k):
    assert np.array_equal(rd.assemble(bg[:, f].numpy(), w, h, ts, 1), frames[f]), "batched frame differs"
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_allgather_assembles_frame(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", RFW_PORT=str(port), RFW_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_shard_geometry_covers_image():
    from rfw_rs_amd import dist as rd
    for (w, h, world, ts) in [(1920, 1080, 8, 64), (200, 136, 3, 32), (64, 64, 1, 64), (65, 9, 2, 8)]:
        owner, slot = rd.slab_index_map(w, h, world, ts)
        g = rd.shard_geometry(w, h, world, ts)
        assert slot.max() < g["slab_elems"] and owner.max() < world
        counts = np.bincount(owner.ravel(), minlength=world)
        assert counts.sum() == w * h
        if g["tiles_total"] >= 4 * world:
            assert counts.max() / max(counts.min(), 1) < 1.6   # round-robin dealing balances pixels


def _bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_plain_bench_command_launches_n_ranks():
    """`python3 bench.py --gpus N` with no launcher: bench.py starts N child ranks itself (before any GPU call; here without a GPU at all:
    --rendezvous-only lets the ranks meet over gloo and report).  One JSON line, from rank 0; every rank a process of its own."""
    import json
    for n in (2, 3):
        r = _bench(["--gpus", str(n), "--rendezvous-only"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["ranks_seen"] == n
        assert sorted(x["rank"] for x in out["ranks"]) == list(range(n)) and len({x["pid"] for x in out["ranks"]}) == n
        assert len({x["launched_by"] for x in out["ranks"]}) == 1 and out["ranks"][0]["launched_by"] is not None


def test_bench_rank_count_follows_gpus_flag_not_stray_environment():
    import json
    # --gpus 1 under a stray WORLD_SIZE: one rank
    r = _bench(["--gpus", "1", "--rendezvous-only"], {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}, drop=())
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["ranks"][0]["rank"] == 0
    # a launcher that started another number of ranks than --gpus says: refused, not reported under the wrong N
    r = _bench(["--gpus", "2", "--rendezvous-only"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port())}, drop=())
    assert r.returncode != 0 and "refusing" in r.stderr
    # under the driver's launcher (python -m torch.distributed.run) the script is one of the launcher's ranks: no second generation of children
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and all(x["launched_by"] is None for x in out["ranks"])


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies takes the job down with a non-zero exit code (and no result line): here every rank dies at once, for want of a device."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "cornell"], {"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_baseline_config_children_are_short_runs_of_the_same_script(monkeypatch):
    """bench.py's headline run reports BASELINE.json's other single-GPU configurations (C2, C3, path-traced C4) as rows of config.modes, each
    from a child run of the same script started BEFORE the parent touches the GPU.  No GPU here: the children are replaced by a fake that
    records their command lines and environments and answers with a bench line."""
    import importlib.util, json, subprocess, sys, types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    spec.loader.exec_module(bench)
    seen = []

    def fake_run(cmd, capture_output, text, timeout, env):
        seen.append((cmd, env))
        line = {"value": 1234.5, "ms_per_step": 0.5, "steps": 200, "cpu_baseline": {"value": 15.0, "cores": 16},
                "config": {"workload": " ".join(cmd[2:]), "mode": "render() per frame, 12 frame slots", "rays_per_frame": 3000000,
                           "timed_frame_equals_oracle": {"all": True}, "modes": {"x": {"Mrays_per_s": 1.0}, "value row": {"Mrays_per_s": 2.0, "is_value": True}}}}
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps(line) + "\n", stderr="")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")   # as the parent sets it for ITS configuration ...
    monkeypatch.setattr(bench, "QUEUES_SET_BY_CALLER", False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    args = types.SimpleNamespace(baseline_steps=200, procedural=False)
    rows = bench.baseline_configs(args)
    assert len(rows) == 3 and all(r["Mrays_per_s"] == 1234.5 and r["timed_frame_equals_oracle"]["all"] for r in rows.values())
    kinds = [" ".join(c[0]) for c in seen]
    assert any("--workload atrium262k" in k for k in kinds) and any("--workload spheres10k" in k for k in kinds)
    assert any("--workload atrium1m --max-path-length 3" in k for k in kinds)
    for cmd, env in seen:
        assert "--no-baseline-configs" in cmd and "--gpus" in cmd and cmd[cmd.index("--steps") + 1] == "200"
        assert "GPU_MAX_HW_QUEUES" not in env and "WORLD_SIZE" not in env   # ... a child chooses its own (C3: 20 slots over 24 queues)
    path_traced = [r for k, r in rows.items() if "config 4" in k][0]
    assert path_traced["modes"] == {"x": 1.0}                             # the child's own secondary modes ride along, its value row does not
    # a child that fails costs its row, not the headline
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout="", stderr="boom"))
    rows = bench.baseline_configs(args)
    assert all("error" in r for r in rows.values())
