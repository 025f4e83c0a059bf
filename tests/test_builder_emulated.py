"""The device builders WITHOUT a device.  tests/emu/wave_emu.h runs the SOURCE TEXT of a kernel with every lane a fiber of its wavefront's OS
thread (an OS thread of its own with RFW_EMU_SANITIZE set) and every cross-lane operation — ballot, shuffles, DPP row operations, wave and
workgroup barriers — a meeting of the lanes; tests/emu/fake_hip stands in for the runtime calls a .hip file's host side makes.  The harnesses
(tests/emu/*_emu.cpp) compare what the kernels build, node for node, with serial restatements: the binned SAH (sah_reference.h), the radix
tree over Morton keys (tlas_fused_emu.cpp).  No GPU, no oracle: logic of the device builders, checked here so that
a change to a kernel is checked BEFORE it first runs on a device — a kernel that never ends, or a tree with a cycle under the traversal
kernels, takes the GPU box down with it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
sys.path.insert(0, EMU)

# capacity (= workgroup size), primitives, seed, kind of boxes (0 random, 1 lattice: equal costs and centroids, 2 two thirds share a centroid,
# 3 flat and collinear, 4 very uneven, 5 pairs at exponentially growing distances: a tree 38 levels deep), largest leaf, traversal cost
CASES = [(256, 1, 1, 0, 8, 1.0), (256, 2, 2, 2, 1, 1.0), (256, 3, 3, 0, 8, 1.0), (256, 17, 4, 1, 8, 1.0), (256, 64, 5, 0, 8, 1.0), (256, 65, 6, 4, 1, 1.0),
         (256, 100, 7, 2, 1, 1.0), (256, 200, 8, 3, 8, 0.5), (256, 256, 9, 0, 8, 1.0), (256, 256, 10, 1, 20, 2.0), (256, 256, 13, 5, 8, 1.0), (512, 300, 11, 0, 8, 1.0),
         (512, 512, 12, 4, 4, 1.0), (512, 512, 14, 5, 8, 1.0), (512, 512, 15, 2, 1, 1.0), (512, 511, 16, 1, 8, 1.0)]
SANITIZE = os.environ.get("RFW_EMU_SANITIZE")   # "thread" | "address,undefined": a thread per lane instead of fibers, built with -fsanitize=...; minutes per test


def compile_harness(tmp, name):
    exe = os.path.join(tmp, name)
    extra = ["-DEMU_THREADS", "-g", "-fsanitize=" + SANITIZE, "-Wno-tsan"] if SANITIZE else []
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-pthread", "-Wno-unknown-pragmas"] + extra + ["-x", "c++", "-I", tmp, "-I", os.path.join(EMU, "fake_hip"),
                        "-I", EMU, "-I", os.path.join(ROOT, "rfw-rs_amd", "csrc"), "-o", exe, os.path.join(EMU, name + ".cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def build_harness(tmp, source=None):
    import extract
    extract.extract(os.path.join(tmp, "k_small_extract.inc"), source)
    return compile_harness(tmp, "k_small_emu")


def run(exe, case):
    """One case of a harness: exit 0, "OK ..." on stdout, and nothing from a sanitizer on stderr (the one report expected of the shipped
    sah_build.hip — k_small's list-full read, DESIGN §10 — aside)."""
    if SANITIZE and len(case) > 1 and isinstance(case[1], int) and max(x for x in case if isinstance(x, int)) > 2000:
        return None   # (a thread per lane: the large cases would take hours)
    r = subprocess.run([exe] + [str(x) for x in case], capture_output=True, text=True, timeout=3000 if SANITIZE else 300)
    assert r.returncode == 0 and r.stdout.rstrip().split("\n")[-1].startswith("OK"), (case, r.stdout[-500:], r.stderr[-1500:])
    reports = [l for l in r.stderr.split("\n") if "WARNING: ThreadSanitizer" in l or "runtime error" in l or "ERROR: AddressSanitizer" in l]
    return r, reports


def run_cases(exe, cases, known_reports=0, leaf_per_primitive=False):
    for case in cases:
        res = run(exe, case)
        if res is None:
            continue
        r, reports = res
        assert len(reports) <= known_reports, (case, r.stderr[-3000:])
        if leaf_per_primitive and case[1] >= 2 and case[4] == 1:   # largest leaf 1: a binary tree with a leaf per primitive
            assert "leaves=%d " % case[1] in r.stdout, (case, r.stdout)


def test_workgroup_phase_of_the_device_builder_equals_its_serial_restatement(tmp_path):
    run_cases(build_harness(str(tmp_path)), CASES, known_reports=1, leaf_per_primitive=True)   # (thread sanitizer: S.list_n)


def test_level_at_a_time_variant_of_the_workgroup_phase_builds_the_same_trees(tmp_path):
    """experiments/segment_levels: the sub-ranges of <= 16 primitives finished one LEVEL per trip by all lanes of the wavefront instead of one
    split after the other.  Not shipped (never run on a device: see its README); the same restatement holds for it."""
    patch = os.path.join(ROOT, "experiments", "segment_levels", "segment_levels.patch")
    src = os.path.join(str(tmp_path), "sah_build.hip")
    with open(os.path.join(ROOT, "rfw-rs_amd", "csrc", "sah_build.hip")) as f, open(src, "w") as g:
        g.write(f.read())
    r = subprocess.run(["patch", "-p3", "-s", src, patch], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "finish_segments_wave" in open(src).read()
    run_cases(build_harness(str(tmp_path), src), CASES, leaf_per_primitive=True)
    # ... and the patched FILE as a whole, up to the size that takes workgroups of 512
    import extract
    extract.whole_file(os.path.join(str(tmp_path), "sah_build.hip"), "sah_build.hip", src)
    run_cases(compile_harness(str(tmp_path), "sah_build_emu"), [(600, 2, 0, 8, 1.0), (700, 5, 1, 8, 1.0, 4), (6000, 7, 4, 8, 1.0), (40000, 3, 1, 8, 1.0)])


def test_one_workgroup_tlas_build_equals_its_serial_restatement(tmp_path):
    """k_tlas_fused (csrc/lbvh.hip; SURVEY §8 a4, the reference's per-frame TLAS rebuild: backends/gpu-rt/src/lib.rs:1576-1615) with its 1024
    threads as OS threads and its 128 KB of LDS as a static array: leaf order, children, parents, every fitted box, the even-depth flags, their
    prefix sum and the 4-wide nodes against tests/emu/tlas_fused_emu.cpp's restatement (stable sort of the Morton keys, the radix tree over
    (key, position) top-down).  Instances, seed, kind (0 scattered, 1 a coarse lattice: many equal keys, 2 all in one place, 3 along a line)."""
    import extract
    extract.extract_tlas(os.path.join(str(tmp_path), "tlas_fused_extract.inc"))
    exe = compile_harness(str(tmp_path), "tlas_fused_emu")
    run_cases(exe, [(2, 1, 0), (3, 1, 0), (16, 8, 0), (17, 1, 0), (33, 9, 2), (777, 3, 2), (1000, 2, 1), (5000, 4, 3), (10000, 5, 0), (16383, 7, 0), (16384, 6, 1), (16384, 8, 2)])


def test_device_sah_builder_as_a_whole_equals_the_serial_binned_sah(tmp_path):
    """csrc/sah_build.hip — the file: host side, level kernels, workgroup phase, emission — compiled against tests/emu/fake_hip (a launch runs its
    workgroups one after the other) builds trees on the CPU; the 4-wide tree it hands to the traversal is compared with the one that follows
    from tests/emu/sah_reference.h.  Primitives, seed, kind (0 scattered, 1 a surface in mesh order, 4 very uneven), largest
    leaf, traversal cost, meshes (> 1: the forest build; 0: one tree and then its REFIT — a16 / f3 — to unrelated triangles, every child box
    checked against the union of what lies below it)."""
    import extract
    extract.whole_file(os.path.join(str(tmp_path), "sah_build.hip"), "sah_build.hip")
    exe = compile_harness(str(tmp_path), "sah_build_emu")
    # 40 000 primitives: workgroups of 512 in phase 2, partition by chunks of blocks and replicated bins on the upper levels of phase 1
    run_cases(exe, [(1, 1, 0, 8, 1.0), (2, 1, 0, 8, 1.0), (100, 1, 0, 8, 1.0), (300, 4, 4, 8, 1.0), (600, 2, 0, 8, 1.0), (700, 5, 1, 8, 1.0, 4), (500, 9, 1, 8, 1.0, 0),
                    (5000, 3, 1, 4, 1.0), (6000, 7, 4, 8, 1.0), (20000, 8, 0, 8, 1.0, 12), (9000, 10, 1, 8, 1.0, 0), (40000, 3, 1, 8, 1.0)], known_reports=2)


def test_lbvh_file_on_the_cpu_passes_its_own_stress_test_and_builds_the_tlas_the_same_both_ways(tmp_path):
    """csrc/lbvh.hip as a whole under the emulator.  `stress`: the library's own rfw_hip_debug_lbvh_stress path (jittered boxes -> lbvh_build with
    the fence-free fit -> the device-side check of every child box and primitive) must report no mismatch; `tlas`: the chain of seventeen launches
    (instance boxes, bounds, Morton keys, radix sort, hierarchy, fit, collapse, gather) and the one workgroup of tlas_build_fused give the same
    instance boxes, leaf order and 4-wide nodes byte for byte — the GPU test of the same name's claim (tests/test_gpu_api.py), without a GPU."""
    import extract
    extract.whole_file(os.path.join(str(tmp_path), "lbvh.hip"), "lbvh.hip")
    exe = compile_harness(str(tmp_path), "lbvh_emu")
    run_cases(exe, [("stress", 1000, 2, 1), ("stress", 20000, 3, 7), ("tlas", 2, 1), ("tlas", 3, 1), ("tlas", 17, 4), ("tlas", 1000, 2), ("tlas", 5000, 3), ("tlas", 16384, 5)])
