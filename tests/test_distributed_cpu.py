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
