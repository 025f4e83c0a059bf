"""The reciprocals behind the kernels' index divisions (csrc/api_frame.cpp index_magic, kernels.hip div_magic) — host arithmetic, no GPU:
n / d == (n * m) >> 32 for EVERY n up to the bound the host asked for, and 0 (= "divide") whenever that cannot be guaranteed."""
import numpy as np
import pytest

from rfw_rs_amd.backend import hip_lib as load_library


def check(lib, d, n_max):
    m = lib.rfw_hip_selftest_index_magic(d, n_max)
    if m == 0:
        return False
    if n_max <= 3_000_000:
        n = np.arange(0, n_max + 1, dtype=np.uint64)
    else:  # the ends, every multiple of d and its neighbours, and a random sample
        k = np.arange(0, n_max // d + 1, dtype=np.uint64) * np.uint64(d)
        rng = np.random.default_rng(d)
        n = np.concatenate([k, k + np.uint64(d - 1), k[1:] - np.uint64(1), rng.integers(0, n_max + 1, 200_000).astype(np.uint64), np.array([n_max], dtype=np.uint64)])
        n = np.unique(n[n <= np.uint64(n_max)])
    assert np.array_equal((n * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), (d, n_max, m)
    return True


def test_every_frame_width_and_tile_count_up_to_4k():
    lib = load_library()
    exact = 0
    for w, h in [(1920, 1080), (1280, 720), (3840, 2160), (640, 480), (1, 1), (7, 5), (33, 17), (4096, 4096), (1000, 1000), (2560, 1440), (8192, 4320)]:
        exact += check(lib, w, w * h)
        for ts in (8, 16, 24, 32, 64, 128):
            tx, ty = -(-w // ts), -(-h // ts)
            exact += check(lib, tx, tx * ty + 4096 * 8)
    assert exact > 40  # (the common cases do get a constant)


def test_random_divisors_and_bounds():
    lib = load_library()
    rng = np.random.default_rng(5)
    for _ in range(300):
        d = int(rng.integers(1, 70000))
        check(lib, d, int(rng.integers(1, 2_000_000)))
    assert lib.rfw_hip_selftest_index_magic(0, 10) == 0 and lib.rfw_hip_selftest_index_magic(1, 10) == 0
    assert lib.rfw_hip_selftest_index_magic(3, 1 << 33) == 0  # (indices are 32-bit: a bound beyond that is refused)
    # a divisor whose round-up reciprocal goes wrong inside the asked range must be refused: 2^32 / 7 rounds up with e = 3 -> wrong from n ~ 2^32 / 3 on
    assert lib.rfw_hip_selftest_index_magic(7, (1 << 32) - 1) == 0
