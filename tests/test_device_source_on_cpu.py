"""The HIP path WITHOUT a GPU: every source file of rfw-rs_amd/csrc compiled for the host (tests/emu/build_emu_lib.py: the ROCm clang as an x86
compiler, tests/emu/fake_hip as the runtime, the kernels under wave_emu.h — a lane is a fiber, a cross-lane operation a meeting of the
wavefront's lanes) into a library with the product's C ABI, and the GPU tests' own assertions run against it through the same Python bindings:
ray queries and accumulated radiance bit for bit against the oracle and against the committed golden vectors.

TEST INFRASTRUCTURE.  The emulated library is built into pytest's temporary directory and loaded only because this file sets RFW_HIP_LIB
for the child processes it starts; nothing of the package knows it, and the product without its HIP library still fails loudly
(tests/test_boundary.py).  What it adds to the `-m gpu` tests (which stay the parity tests proper): the kernels' SOURCE — traversal of both
kinds, shading, queues, builders, TLAS, refit, textures — is held to the oracle on every CPU run, by a different compiler, before anything
reaches a device.  What it cannot see: the device compiler, timing, v_rcp_f32's last bit (conservative box tests only), memory ordering below
a workgroup."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang as a host compiler")


@pytest.fixture(scope="module")
def emulated_library(tmp_path_factory):
    import build_emu_lib
    return build_emu_lib.build(str(tmp_path_factory.mktemp("emulated_hip")))


def run_gpu_tests(lib, args, timeout=1500):
    env = dict(os.environ, RFW_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    return tail


def test_golden_vectors_from_the_emulated_kernels(emulated_library):
    """tests/test_golden.py::test_hip_reproduces_golden — cornell / soup / gallery / skinned, primary + shadow and path traced, blue noise — on the
    emulated library: the accumulators equal the committed vectors bit for bit."""
    out = run_gpu_tests(emulated_library, ["tests/test_golden.py", "-k", "hip_reproduces_golden"])
    assert "6 passed" in out, out


def test_parity_tests_on_the_emulated_kernels(emulated_library):
    """The parity tests proper (tests/test_gpu_parity.py) at the sizes they run on the device: closest and any hit against the oracle's tree and
    against brute force, radiance of four scenes, ray counts and queue sizes, the accumulator's reset, animated instances, every builder."""
    out = run_gpu_tests(emulated_library, ["tests/test_gpu_parity.py", "-k",
                                           "closest_hit_bit_exact or any_hit_exact or edge_cases or radiance_matches_oracle or counts_and_queues or accumulation_reset or "
                                           "animated_instances_match_oracle or builders_give_identical_answers"])
    assert " passed" in out and "failed" not in out, out
