"""Spatial splits (round 5; MeshHost in csrc/api_internal.h): what set_3d_mesh computes on the host, through rfw_hip_selftest_splits — no GPU.

A triangle whose box wastes more than `spatial_splits` x the area of the mesh's box is referenced several times, each reference with the
tight box of one part of the triangle.  Checked here: which triangles are split, that the parts' boxes cover their triangle and stay
inside its box, the bookkeeping of duplicates, the budget, and that the result does not depend on the number of copy threads.  That the
IMAGE does not depend on it is checked on the device (tests/test_gpu_parity.py: test_spatial_splits_never_change_the_image)."""
import ctypes as C

import numpy as np
import pytest

from rfw_rs_amd import backend, pod


def mesh(n_small=12000, seed=3, big=True):
    """A 20 x 10 x 20 room filled with small triangles, two wall triangles across it and a long diagonal sliver."""
    rng = np.random.default_rng(seed)
    c = rng.uniform([0, 0, 0], [20, 10, 20], (n_small, 1, 3))
    v = (c + rng.uniform(-0.05, 0.05, (n_small, 3, 3))).astype(np.float32)
    if big:
        walls = np.array([[[0, 0, 0], [20, 0, 0], [20, 10, 0]], [[0, 0, 0], [20, 10, 0], [0, 10, 0]],   # a wall of two triangles
                          [[0, 0, 20], [20, 10, 0.5], [20, 9.9, 0.5]]], dtype=np.float32)                  # a sliver through the room
        v = np.concatenate([v[:1000], walls[:2], v[1000:], walls[2:]])
    tris = (pod.RTTriangle * len(v))()
    for i, t in enumerate(v):
        for k, name in enumerate(("vertex0", "vertex1", "vertex2")):
            f = getattr(tris[i], name)
            f.x, f.y, f.z = map(float, t[k])
        n = np.cross(t[1] - t[0], t[2] - t[0])
        n = n / max(np.linalg.norm(n), 1e-30)
        tris[i].normal.x, tris[i].normal.y, tris[i].normal.z = map(float, n)
        tris[i].v0 = 0.25  # (the texture coordinate the duplicates overwrite with their original's index)
    return v, tris


def run(tris, tau, threads):
    lib = backend.hip_lib()
    n = len(tris)
    pieces = np.zeros((n, 7), dtype=np.float32)
    dup = np.zeros(n, dtype=np.uint32)
    npc = C.c_uint32(0)
    refs = lib.rfw_hip_selftest_splits(C.addressof(tris), n, tau, threads, pieces.ctypes.data, n, dup.ctypes.data, n, C.byref(npc))
    assert refs >= n
    return int(refs), pieces[:npc.value].copy(), dup[:refs - n].copy()


def test_only_the_wasteful_triangles_are_split_and_their_parts_cover_them():
    v, tris = mesh()
    n = len(v)
    refs, pieces, dup = run(tris, 2e-4, 4)
    big = {1000, 1001, n - 1}
    idx = pieces[:, 0].astype(np.int64)
    owners = np.where(idx < n, idx, dup[np.clip(idx - n, 0, max(len(dup) - 1, 0))])
    assert set(owners.tolist()) == big                       # nothing else is worth a reference of its own
    assert refs == n + len(dup) and refs - n <= min(n // 128 + 64, 1024)  # inside the budget the arrays were allocated with
    assert set(dup.tolist()) <= big
    assert sorted(idx[idx >= n].tolist()) == list(range(n, refs))  # every duplicate has exactly one box
    rng = np.random.default_rng(0)
    for t in big:
        mine = pieces[owners == t]
        assert (mine[:, 0] == t).sum() == 1                   # the triangle's own entry is one of its references
        assert 2 <= len(mine) <= 64
        lo, hi = v[t].min(0), v[t].max(0)
        assert (mine[:, 1:4] >= lo - 1e-4).all() and (mine[:, 4:7] <= hi + 1e-4).all()
        # points of the triangle lie in the box of at least one part (the device pads every box by 1e-4 + 4e-6 |c|)
        b = rng.dirichlet((1, 1, 1), 4000).astype(np.float64)
        p = b @ v[t].astype(np.float64)
        inside = ((p[:, None, :] >= mine[None, :, 1:4] - 2e-4) & (p[:, None, :] <= mine[None, :, 4:7] + 2e-4)).all(-1).any(-1)
        assert inside.all()
        # ... and the parts are what they are for: together their boxes have far less surface than the triangle's one box
        area = lambda l, h: 2 * ((h - l)[..., 0] * (h - l)[..., 1] + (h - l)[..., 1] * (h - l)[..., 2] + (h - l)[..., 2] * (h - l)[..., 0])
        if t != n - 1:
            continue
        assert area(mine[:, 1:4], mine[:, 4:7]).sum() < 0.5 * area(lo, hi)


def test_result_does_not_depend_on_the_thread_count_and_zero_switches_it_off():
    _, tris = mesh(n_small=60000)  # (large enough for several copy threads: 4 MB per thread)
    a = run(tris, 2e-4, 1)
    for threads in (2, 8):
        b = run(tris, 2e-4, threads)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    refs, pieces, dup = run(tris, 0.0, 4)
    assert refs == len(tris) and len(pieces) == 0 and len(dup) == 0
    _, tiny = mesh(n_small=3000)                              # below 8192 triangles a mesh is neither scanned nor split, whatever it holds
    refs, pieces, dup = run(tiny, 2e-4, 4)
    assert refs == len(tiny) and len(pieces) == 0
    _, small = mesh(n_small=12000, big=False)                # nothing wasteful: nothing split
    refs, pieces, dup = run(small, 2e-4, 4)
    assert refs == len(small) and len(pieces) == 0


def test_budget_holds_when_every_triangle_is_wasteful():
    """Long diagonal slivers only: the most wasteful parts are cut first until the budget of duplicates (n / 128 + 64) is spent."""
    rng = np.random.default_rng(5)
    n = 10000
    a = rng.uniform(0, 10, (n, 1, 3))
    v = np.concatenate([a, a + rng.uniform(5, 10, (n, 1, 3)), a + rng.uniform(5, 10, (n, 1, 3)) * [[1, 0.01, 1]]], axis=1).astype(np.float32)
    tris = (pod.RTTriangle * n)()
    for i, t in enumerate(v):
        for k, name in enumerate(("vertex0", "vertex1", "vertex2")):
            f = getattr(tris[i], name)
            f.x, f.y, f.z = map(float, t[k])
    refs, pieces, dup = run(tris, 1e-6, 4)
    assert refs - n == min(n // 128 + 64, 1024)
    assert len(pieces) >= refs - n


def test_vertices_on_successive_mid_planes_never_leave_a_part_uncovered():
    """a vertex that lies exactly ON a cutting plane is kept by both halves, so a part can collect more vertices than a clipped
    triangle normally has; a part whose polygon would not fit is left uncut instead of losing vertices (a box that does not cover its part
    of the triangle means missed hits).  Triangles with vertices on the dyadic mid-planes of their own boxes: the union of the parts' boxes
    covers every point of the triangle."""
    rng = np.random.default_rng(11)
    n_small = 9000
    c = rng.uniform([0, 0, 0], [16, 16, 16], (n_small, 1, 3))
    v = (c + rng.uniform(-0.02, 0.02, (n_small, 3, 3))).astype(np.float32)
    big = np.array([[[0, 0, 0], [16, 8, 4], [8, 16, 12]],      # every coordinate a multiple of the successive mid-planes of its box
                    [[0, 8, 16], [16, 8, 0], [8, 0, 8]],
                    [[4, 4, 4], [12, 12, 4], [8, 8, 16]],
                    [[0, 0, 8], [16, 16, 8], [16, 0, 8.0001]]], dtype=np.float32)
    v = np.concatenate([v, big])
    tris = (pod.RTTriangle * len(v))()
    for i, t in enumerate(v):
        for k, name in enumerate(("vertex0", "vertex1", "vertex2")):
            f = getattr(tris[i], name)
            f.x, f.y, f.z = map(float, t[k])
    n = len(v)
    refs, pieces, dup = run(tris, 1e-5, 4)
    idx = pieces[:, 0].astype(np.int64)
    owners = np.where(idx < n, idx, dup[np.clip(idx - n, 0, max(len(dup) - 1, 0))])
    split = set(owners.tolist())
    assert split & set(range(n_small, n))                      # the big ones are cut
    for t in sorted(split):
        mine = pieces[owners == t]
        b = rng.dirichlet((1, 1, 1), 6000).astype(np.float64)
        b = np.concatenate([b, np.eye(3), [[0.5, 0.5, 0], [0, 0.5, 0.5], [0.5, 0, 0.5]]])   # corners and edge mid-points too
        p = b @ v[t].astype(np.float64)
        inside = ((p[:, None, :] >= mine[None, :, 1:4] - 1e-4) & (p[:, None, :] <= mine[None, :, 4:7] + 1e-4)).all(-1).any(-1)
        assert inside.all(), (t, p[~inside][:3])
