"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): the golden cases that exercise every part of it —
Cornell (path length 3), the instanced soup (TLAS + BLAS, blue-noise tables), the textured gallery (texture array, normal maps, skybox), the
skinned scene — rendered in a child process with libasan preloaded; any report aborts the child.  The sanitized build keeps the arithmetic
contract (-ffp-contract=off), so its images must be the committed goldens bit for bit as well."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import oracle.bindings as ob
ob.ORACLE_LIB = sys.argv[2]          # every Oracle() of this process loads the sanitized build
from make_golden import CASES, run_case
for name in sys.argv[3:]:
    acc, hits, st = run_case(CASES[name])
    g = np.load(os.path.join(sys.argv[1], "tests", "golden", name + ".npz"))
    assert np.array_equal(acc.view(np.uint32), g["acc"].view(np.uint32)), name
    assert np.array_equal(hits["tri"], g["hit_tri"]), name
    print("clean", name, flush=True)
"""


def test_oracle_renders_the_goldens_clean_under_asan_and_ubsan(tmp_path):
    cxx = os.environ.get("CXX", "g++")
    libasan = subprocess.run([cxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan on this machine")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    cases = ["cornell_path3_4spp", "soup_instanced_path3_2spp", "gallery_textured_path3_2spp", "skinned_pose07_path3_2spp", "soup_blue_noise_path3_3spp"]
    # (detect_leaks=0: the interpreter itself is not leak-clean; everything else — heap / stack / global overruns, use after free, UB — is fatal)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(script), ROOT, lib] + cases, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count("clean ") == len(cases)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
