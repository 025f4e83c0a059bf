"""include/rfw_detmath.h against libm (numpy float64): the shared elementary functions must be accurate,
since oracle and kernels both use them (a shared bug would be invisible to parity tests)."""
import numpy as np

from oracle.bindings import detmath


def max_ulp(got, ref64):
    ref32 = ref64.astype(np.float32)
    sp = np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30))).astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - ref64) / sp))


def test_sincos():
    x = np.linspace(0.0, 7.1, 400001, dtype=np.float32)  # path range: TWOPI*r and blade angles <= 10*pi/4.5
    x64 = x.astype(np.float64)
    assert max_ulp(detmath("sin", x), np.sin(x64)) <= 2.0 or np.max(np.abs(detmath("sin", x) - np.sin(x64))) < 2e-7
    assert max_ulp(detmath("cos", x), np.cos(x64)) <= 2.0 or np.max(np.abs(detmath("cos", x) - np.cos(x64))) < 2e-7
    xs = np.float32([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi])
    assert np.allclose(detmath("sin", xs), np.sin(xs.astype(np.float64)), atol=2e-7)
    s, c = detmath("sin", x), detmath("cos", x)
    assert np.max(np.abs(s * s + c * c - 1.0)) < 5e-7
    xn = -x
    assert np.array_equal(detmath("sin", xn), -s) and np.array_equal(detmath("cos", xn), c)


def test_log_exp():
    x = np.geomspace(1e-12, 1e6, 200001).astype(np.float32)
    assert max_ulp(detmath("log", x), np.log(x.astype(np.float64))) <= 2.0
    assert max_ulp(detmath("log2", x), np.log2(x.astype(np.float64))) <= 4.0
    x = np.linspace(-30, 20, 200001, dtype=np.float32)
    assert max_ulp(detmath("exp", x), np.exp(x.astype(np.float64))) <= 2.0
    assert detmath("exp", np.float32([0.0]))[0] == 1.0
    assert detmath("exp", np.float32([-200.0]))[0] == 0.0 and np.isinf(detmath("exp", np.float32([100.0]))[0])


def test_inverse_trig():
    x = np.linspace(-1, 1, 200001, dtype=np.float32)
    assert np.max(np.abs(detmath("acos", x) - np.arccos(x.astype(np.float64)))) < 5e-7
    assert np.max(np.abs(detmath("asin", x) - np.arcsin(x.astype(np.float64)))) < 5e-7
    rng = np.random.default_rng(3)
    y, xx = rng.normal(size=100000).astype(np.float32), rng.normal(size=100000).astype(np.float32)
    assert np.max(np.abs(detmath("atan2", y, xx) - np.arctan2(y.astype(np.float64), xx.astype(np.float64)))) < 1e-6
