"""The drop-in boundary: POD layouts, header consistency, exported symbols (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

from conftest import ROOT
from rfw_rs_amd import EXPORTS, HIP_LIB, pod


def test_pod_sizes_match_reference_layout_test():
    # restates backends/metal/src/lib.rs:270-348 (size_of Rust POD == size_of C POD) with the sizes of SURVEY.md Appendix A
    for t, n in pod.EXPECTED_SIZES.items():
        assert C.sizeof(t) == n
    assert pod.RTTriangle.tangent0.offset == 112 and pod.RTTriangle.light_id.offset == 160 and pod.RTTriangle.area.offset == 172
    assert pod.CameraView3D.custom0.offset == 96 and pod.CameraView3D.lens_size.offset == 60
    assert pod.DeviceMaterial.flags.offset == 64 and pod.AreaLight.radiance.offset == 64 and pod.SpotLight.energy.offset == 44


def test_headers_compile_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "rfw_hip.h"\n#include "rfw_detmath.h"\nint main(void){return (int)rfw_sinf(0.0f);}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rfw_hip.h")).read()
    declared = sorted(set(re.findall(r"RFW_HIP_API[^;(]*?\b(rfw_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 30
    assert sorted(EXPORTS) == declared, set(EXPORTS) ^ set(declared)
    lib = C.CDLL(HIP_LIB)  # loads without a GPU; no compute call is made here
    for name in declared:
        assert hasattr(lib, name), name
    lib.rfw_hip_abi_version.restype = C.c_uint32
    assert lib.rfw_hip_abi_version() == 2


def test_create_fails_loudly_without_gpu():
    from conftest import has_gpu
    from rfw_rs_amd import BackendError, HipBackend
    if has_gpu():
        return
    try:
        HipBackend.init(16, 16)
    except BackendError as e:
        assert "no HIP device" in str(e) or "hip" in str(e).lower()
    else:
        raise AssertionError("create must fail without a HIP device: there is no CPU fallback")


def test_product_never_touches_oracle():
    # the oracle is test infrastructure: nothing under rfw-rs_amd/ may include, import or link it
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rfw-rs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r'oracle/|liboracle|import oracle|from oracle', txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    out = subprocess.run(["ldd", HIP_LIB], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out


def test_host_bvh_builder_and_quantiser_selftest():
    """The product's CPU-side builder (binned SAH -> 4-wide collapse) and the 8-bit node quantiser, without a GPU."""
    import numpy as np
    from rfw_rs_amd import hip_lib
    lib = hip_lib()
    rng = np.random.default_rng(3)
    for n, leaf, threads in [(0, 4, 1), (1, 4, 1), (2, 4, 1), (7, 1, 2), (5000, 4, 4), (20000, 8, 8)]:
        c = rng.uniform(-50, 50, size=(n, 3)).astype(np.float32)
        e = rng.uniform(0.0, 2.0, size=(n, 3)).astype(np.float32)
        if n > 10:
            e[:10] = 0.0                      # degenerate (point) boxes
            c[10:20] = c[10]                  # coincident centroids
        boxes = np.ascontiguousarray(np.concatenate([c - e, c + e], axis=1))
        nodes = C.c_uint32(0)
        err = lib.rfw_hip_selftest_bvh(boxes.ctypes.data, n, leaf, threads, C.byref(nodes))
        assert err == 0, (n, leaf, err)
        assert nodes.value >= 1
