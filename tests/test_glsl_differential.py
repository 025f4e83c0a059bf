"""The oracle's hand-written twins of the reference's pure shader functions against a THIRD build of the same functions made from the
reference's own text: oracle/make_glsl_ref.py reads backends/gpu-rt/shaders/{utils,random,structs,disney,intersection}.glsl where they lie
(nothing is copied; the test skips where /root/reference is absent — the GPU box), rewrites qualifiers / literal suffixes / `.xyz` and
compiles them as C++ behind oracle/glsl_shim.h.  Every function is then evaluated on 10^5 seeded inputs on both sides and compared BIT FOR
BIT.  A transliteration slip that sits in both of our textual twins (oracle/oracle.cpp and csrc/shade_device.h agree bit for bit on the
device) shows up here; what this cannot show is what a real GLSL compiler and GPU would do with the built-ins GLSL leaves open — the
stand-in gives them the meanings oracle/glsl.h pins — so DESIGN.md keeps "parity unpinned" (VERDICT r03, next #6)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 100_000


@pytest.fixture(scope="module")
def libs():
    from oracle import make_glsl_ref
    if not make_glsl_ref.available():
        pytest.skip("the reference checkout (/root/reference) is not on this machine")
    ref = C.CDLL(make_glsl_ref.build())
    from oracle import bindings
    orc = bindings.lib()
    for l, names in ((ref, ("glslref_eval_shading", "glslref_twin")), (orc, ("orc_glsl_twin",))):
        for n in names:
            getattr(l, n).argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
            getattr(l, n).restype = C.c_int
    return ref, orc


def same_bits(a, b):
    """bit equality, any NaN equal to any NaN"""
    au, bu = a.view(np.uint32), b.view(np.uint32)
    return (au == bu) | (np.isnan(a) & np.isnan(b))


def twin(libs, op, q):
    ref, orc = libs
    q = np.ascontiguousarray(q, np.float32)
    a = np.zeros((len(q), 24), np.float32)
    b = np.zeros((len(q), 24), np.float32)
    assert ref.glslref_twin(op, len(q), q.ctypes.data, a.ctypes.data) == 0
    assert orc.orc_glsl_twin(op, len(q), q.ctypes.data, b.ctypes.data) == 0
    return a, b


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_triangle_tests_agree(libs):
    rng = np.random.default_rng(1)
    q = np.zeros((N, 32), np.float32)
    q[:, 0:9] = rng.uniform(-2, 2, (N, 9))
    e1, e2 = q[:, 3:6] - q[:, 0:3], q[:, 6:9] - q[:, 0:3]
    q[:, 9:12] = np.cross(e1, e2) * rng.uniform(0.2, 3.0, (N, 1)).astype(np.float32)  # any normal along the geometric one (uv are scaled by 1 / |gn|^2)
    q[:, 12:15] = rng.uniform(-4, 4, (N, 3))
    target = q[:, 0:3] + e1 * rng.uniform(-0.2, 1.2, (N, 1)) * 0.5 + e2 * rng.uniform(-0.2, 1.2, (N, 1)) * 0.5  # aimed at (and a little beside) the triangle
    d = target - q[:, 12:15]
    q[:, 15:18] = d / np.linalg.norm(d, axis=1, keepdims=True)
    q[: N // 50, 15:18] = e1[: N // 50] / np.linalg.norm(e1[: N // 50], axis=1, keepdims=True)   # rays in the triangle's plane: the |a| < 1e-4 reject
    q[:, 18] = 1e-4
    q[:, 19] = rng.choice(np.array([1e26, 3.0, 6.0], np.float32), N)
    for op in (10, 11):
        a, b = twin(libs, op, q)
        assert same_bits(a, b).all(), op
        assert 0.2 < a[:, 0].mean() < 0.9   # both outcomes are exercised


def test_mnode_test_and_its_sorting_network_agree(libs):
    rng = np.random.default_rng(2)
    q = np.zeros((N, 32), np.float32)
    lo = rng.uniform(-3, 3, (N, 3, 4)).astype(np.float32)
    hi = lo + rng.uniform(0.0, 2.5, (N, 3, 4)).astype(np.float32)
    for a in range(3):
        q[:, 8 * a:8 * a + 4] = lo[:, a]
        q[:, 8 * a + 4:8 * a + 8] = hi[:, a]
    q[:, 24:27] = rng.uniform(-4, 4, (N, 3))
    c0 = ((lo + hi) * np.float32(0.5))[:, :, 0] + rng.normal(scale=0.7, size=(N, 3)).astype(np.float32)   # aimed near the first child's box
    d = c0 - q[:, 24:27]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[: N // 20, 0] = 0.0   # axis-parallel rays: 1 / 0 = inf in the slab test
    with np.errstate(divide="ignore"):
        q[:, 27:30] = np.float32(1.0) / d
    q[:, 30] = rng.choice(np.array([1e26, 2.0, 5.0], np.float32), N)
    a, b = twin(libs, 12, q)
    assert same_bits(a, b).all()
    assert 0.3 < a[:, 0].mean() < 0.99


def test_small_functions_agree(libs):
    rng = np.random.default_rng(3)
    # safe_origin: small and large coordinates, both sides of the 1 / 32 switch
    q = np.zeros((N, 32), np.float32)
    q[:, 0:3] = rng.uniform(-1, 1, (N, 3)) * rng.choice(np.array([0.01, 0.05, 1.0, 50.0], np.float32), (N, 1))
    q[:, 3:6] = unit(rng, N); q[:, 6:9] = unit(rng, N); q[:, 9] = 1e-4
    a, b = twin(libs, 13, q); assert same_bits(a, b).all()
    # PackNormal (N.z > -1)
    q = np.zeros((N, 32), np.float32); q[:, 0:3] = unit(rng, N)
    a, b = twin(libs, 14, q); assert same_bits(a, b).all()
    # hemisphere sampling, clamp, RNG
    q = np.zeros((N, 32), np.float32); q[:, 0:2] = rng.uniform(0, 1, (N, 2))
    a, b = twin(libs, 16, q); assert same_bits(a, b).all()
    q = np.zeros((N, 32), np.float32); q[:, 0:3] = rng.uniform(0, 30, (N, 3)); q[:, 3] = 10.0
    a, b = twin(libs, 17, q); assert same_bits(a, b).all()
    q = np.zeros((N, 32), np.float32); q[:, 0] = rng.integers(1, 2**32, N, dtype=np.uint64).astype(np.uint32).view(np.float32)
    a, b = twin(libs, 18, q); assert (a.view(np.uint32) == b.view(np.uint32)).all()
    # the BSDF's building blocks
    q = np.zeros((N, 32), np.float32)
    q[:, 0] = rng.uniform(0, 1, N); q[:, 1] = rng.uniform(0.01, 2.0, N)
    q[:, 2:5] = unit(rng, N); q[:, 5:8] = unit(rng, N); q[:, 8] = rng.uniform(0.4, 2.5, N)
    a, b = twin(libs, 20, q); assert same_bits(a, b).all()
    # material unpacking
    q = np.zeros((N, 32), np.float32); q[:, 0:9] = rng.uniform(0, 2, (N, 9))
    q[:, 9:13] = rng.integers(0, 2**32, (N, 4), dtype=np.uint64).astype(np.uint32).view(np.float32)
    a, b = twin(libs, 21, q); assert same_bits(a, b).all()


def shading_cases(rng, n):
    """48 floats per case: the layout of rfw_hip_debug_eval_shading (a 96-byte material, N, wo, wi, T, B, t, backfacing, r3, r4, area)"""
    q = np.zeros((n, 48), np.float32)
    q[:, 0:3] = rng.uniform(0.02, 1.0, (n, 3)); q[:, 4:7] = rng.uniform(0, 1, (n, 3)); q[:, 8:11] = rng.uniform(0, 1, (n, 3))
    par = rng.integers(0, 256, (n, 16), dtype=np.uint64).astype(np.uint8)
    par[rng.uniform(size=n) < 0.5, 10] = 0        # transmission = 0 for half the cases (the opaque branch)
    q[:, 12:16] = par.view(np.uint32).view(np.float32)
    Nn = unit(rng, n)
    T = np.cross(Nn, unit(rng, n)); T /= np.linalg.norm(T, axis=1, keepdims=True)
    B = np.cross(Nn, T)
    def around(nrm):   # directions mostly in the upper hemisphere, some below
        v = unit(rng, n); s = np.sign((v * nrm).sum(1, keepdims=True)); f = np.where(rng.uniform(size=(n, 1)) < 0.85, s, -s)
        return (v * f).astype(np.float32)
    q[:, 24:27] = Nn; q[:, 27:30] = around(Nn); q[:, 30:33] = around(Nn); q[:, 33:36] = T.astype(np.float32); q[:, 36:39] = B.astype(np.float32)
    q[:, 39] = rng.uniform(0, 5, n); q[:, 40] = rng.integers(0, 2, n); q[:, 41:43] = rng.uniform(0, 1, (n, 2)); q[:, 43] = rng.uniform(0.1, 4, n)
    return q


@pytest.mark.parametrize("op,name", [(0, "BSDFEval"), (1, "BSDFPdf"), (2, "BSDFSample")])
def test_disney_bsdf_agrees(libs, op, name):
    ref, _ = libs
    from oracle.bindings import Oracle
    q = shading_cases(np.random.default_rng(10 + op), N)
    a = np.zeros((N, 12), np.float32)
    assert ref.glslref_eval_shading(op, N, q.ctypes.data, a.ctypes.data) == 0
    b = Oracle(8, 8).eval_shading(op, q)
    ok = same_bits(a, np.asarray(b, np.float32).reshape(N, 12))
    assert ok.all(), (name, int((~ok).any(axis=1).sum()), q[np.argmax((~ok).any(axis=1))].tolist())
    assert np.isfinite(a[:, 0]).mean() > 0.9
