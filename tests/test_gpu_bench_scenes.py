"""The BASELINE.json configurations themselves under the oracle (VERDICT r01 "next" #1): the bench scenes — the 262 267- and the
1 048 576-triangle atrium, the 10 000 animated instances — traced by the HIP path through the C ABI and by the oracle on identical
inputs; ray queries and accumulators must be bit-identical (the north star's 1e-4 relative L2 is asserted too).  The oracle runs on
the GPU box's host cores (it finishes a 480x270 path-traced frame of the 1 M-triangle scene in well under a second there)."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-4
THREADS = max(8, os.cpu_count() or 8)


def scene_rays(n, seed):
    """Incoherent rays inside the atrium's bounds (|x| < 15, 0 < y < 12, |z| < 6)."""
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-14, 14, n), rng.uniform(0.2, 11.5, n), rng.uniform(-5.5, 5.5, n)], axis=1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def assert_hits_equal(g, r):
    assert np.array_equal(g["inst"], r["inst"]) and np.array_equal(g["tri"], r["tri"])
    hit = r["inst"] >= 0
    for f in ("t", "u", "v"):
        assert np.array_equal(g[f][hit].view(np.uint32), r[f][hit].view(np.uint32)), f
    return hit


@pytest.fixture(scope="module", params=[262267, 1048576], ids=["C2_atrium262k", "C4_atrium1m"])
def atrium(request):
    """One scene + one synchronized oracle per triangle count, shared by the tests below (the oracle's BVH build is the slow part)."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import Scene
    w, h = 480, 270
    scene = Scene().build("atrium", request.param, 0, 0.0, 0xC0FFEE)   # exactly bench.py's scene
    scene.set_aspect(w / h)
    orc = Oracle(w, h, threads=THREADS)
    scene.sync(orc)
    return request.param, scene, orc, w, h


def test_bench_scene_ray_queries_match_oracle(atrium):
    from rfw_rs_amd import HipBackend
    tris, scene, orc, w, h = atrium
    be = HipBackend.init(w, h, 1.0)
    scene.mark_all_changed(); scene.sync(be)
    assert be.scene_stats()["triangles"] == orc.stats()["n_tris"] and abs(orc.stats()["n_tris"] - tris) < 0.01 * tris   # the generator lands within a fraction of a percent of its target
    o, d = scene_rays(60000, 5)
    po, pd = orc.primary_rays(scene.view(w, h), 0)           # the frame's own (coherent) primary rays as well
    o, d = np.concatenate([o, po[::2]]), np.concatenate([d, pd[::2]])
    g = be.intersect(o, d)
    hit = assert_hits_equal(g, orc.intersect(o, d))
    assert 0.9 < hit.mean() <= 1.0                            # a closed room: almost every ray lands
    tmax = np.random.default_rng(9).uniform(0.05, 20.0, len(o)).astype(np.float32)
    assert np.array_equal(be.occludes(o, d, tmax), orc.occludes(o, d, tmax))
    be.close()


@pytest.mark.parametrize("mpl,spp", [(1, 1), (3, 4)], ids=["primary_shadow_1spp", "path3_4spp"])
def test_bench_scene_radiance_matches_oracle(atrium, mpl, spp):
    """C2 / the headline workload (primary + shadow, 1 spp) and C4 (max path length 3, 4 spp, NEE) at 480 x 270."""
    from rfw_rs_amd import HipBackend
    tris, scene, orc, w, h = atrium
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=mpl)
    scene.mark_all_changed(); scene.sync(be)
    orc.set_option("max_path_length", mpl)
    orc.reset()
    for _ in range(spp):
        be.render(view); orc.render(view)
    ga, ra = be.accumulator(), orc.accumulator()
    assert np.isfinite(ga).all() and ra[..., :3].max() > 0
    assert rel_l2(ga, ra) <= TOL
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), f"{(ga != ra).sum()} differing floats"
    assert np.array_equal(be.framebuffer().view(np.uint32), orc.framebuffer().view(np.uint32))
    s = be.frame_stats()
    assert s["sample_count"] == spp and s["primary_rays"] == w * h and s["shadow_rays"] > 0.2 * w * h
    be.close()


def test_c4_full_size_frame_matches_oracle():
    """BASELINE config C4 at FULL size — 1920 x 1080, 1 048 568 triangles, max path length 3 with NEE — one sample, bit for bit;
    and the headline workload (primary + shadow) of the same scene, the frame bench.py times."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 1920, 1080
    scene = Scene().build("atrium", 1048576, 0, 0.0, 0xC0FFEE)
    view = scene.view(w, h)
    orc = Oracle(w, h, threads=THREADS)
    scene.sync(orc)
    for mpl in (1, 3):
        be = HipBackend.init(w, h, 1.0, max_path_length=mpl)
        scene.mark_all_changed(); scene.sync(be)
        orc.set_option("max_path_length", mpl)
        orc.reset()
        be.render(view); orc.render(view)
        ga, ra = be.accumulator(), orc.accumulator()
        assert rel_l2(ga, ra) <= TOL
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), (mpl, int((ga != ra).sum()))
        s, so = be.frame_stats(), orc.stats()
        assert s["primary_rays"] == w * h
        be.close()


def test_c3_ten_thousand_animated_instances_match_oracle():
    """BASELINE config C3 at size: the 262 267-triangle atrium + 100 x 100 icosphere instances, every instance moved every frame
    (set_3d_instances + synchronize + render: TLAS rebuilt on the device), three frames at 320 x 180 — through one instance with frame
    slots, as bench.py drives it."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    w, h = 320, 180
    scene = Scene().build("atrium", 262267, 0, 0.0, 0xC0FFEE).build("spheres", 100, 100, 0.28)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    be = HipBackend.init(w, h, 1.0, max_path_length=1, frames_in_flight=3)
    orc = Oracle(w, h, threads=THREADS, max_path_length=1)
    o, d = scene_rays(30000, 3)
    o[:, 1] = np.abs(o[:, 1]) * 0.25 + 0.3                    # among the spheres
    for frame in range(3):
        scene.animate(frame / 60.0)                            # examples/animated/src/main.rs:197-219, t = frame / 60
        scene.sync(be)
        scene.mark_all_changed(); scene.sync(orc)
        orc.reset()
        be.render(view); orc.render(view)
        assert be.frame_stats()["sample_count"] == 1
        ga, ra = be.accumulator(), orc.accumulator()
        assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32)), frame
        g = be.intersect(o, d)
        hit = assert_hits_equal(g, orc.intersect(o, d))
        assert (g["inst"][hit] > 0).sum() > 500               # rays do land on sphere instances (instance 0 is the atrium)
    st = be.scene_stats()
    assert st["instances"] == 10001 and st["tlas_nodes"] > 2000
    be.close()


@pytest.mark.parametrize("tris,seed", [(60000, 77), (262267, 78)])
def test_large_scene_soak_cases(tris, seed):
    """The two large cases of tests/soak_gpu.py as collected tests: other seeds of the atrium generator, incoherent rays, one path-traced
    frame through frame slots."""
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend, Scene
    rng = np.random.default_rng(seed)
    scene = Scene().build("atrium", tris, 0, 0.0, seed)
    w, h = 320, 180
    scene.set_aspect(w / h)
    be = HipBackend.init(w, h, 1.0, max_path_length=3, frames_in_flight=2)
    orc = Oracle(w, h, threads=THREADS, max_path_length=3)
    scene.sync(be); scene.mark_all_changed(); scene.sync(orc)
    o = rng.uniform(-8, 8, (50000, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.2
    d = rng.normal(size=(50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert_hits_equal(be.intersect(o, d), orc.intersect(o, d))
    view = scene.view(w, h)
    be.render(view); orc.render(view)
    assert np.array_equal(be.accumulator().view(np.uint32), orc.accumulator().view(np.uint32))
    be.close()
