#!/usr/bin/env python3
"""oracle/make_glsl_ref.py — TEST INFRASTRUCTURE (build container only).

Builds oracle/_ref/libglsl_ref.so from the reference's OWN shader text: the pure functions of
    /root/reference/backends/gpu-rt/shaders/{utils,random,structs,disney,intersection}.glsl
are read IN PLACE, rewritten textually just enough to be C++ (the rewrites are listed below; no arithmetic is touched), and compiled behind
oracle/glsl_shim.h with oracle/glsl_ref_wrap.cpp as the C entry points.  Nothing of the reference is copied into the repository: the
rewritten text exists in a temporary directory only while the compiler runs, and the library lives under oracle/_ref/ (git-ignored), and tests/test_glsl_differential.py skips where /root/reference is
absent.  It is a stand-in RUNTIME for GLSL — it pins nothing about the reference's results on a GPU (DESIGN.md keeps "parity unpinned") — but
it is a third build of the same functions, made from the reference's text rather than from ours, against which the oracle's hand-written
twins are compared bit for bit.

Rewrites:  #include lines dropped (the files are concatenated in dependency order) · `inout T x` / `out T x` -> `T& x` · unsuffixed float
literals get an `f` (GLSL literals are 32-bit; C++ would compute in double) · `.xyz` -> `.xyz_()` · `sign(` -> `gl_sign_f(` (utils.glsl
also names a variable `sign`) · `uint(` / `int(` / `float(` conversions -> to_uint( / to_int( / to_float( (saturating, as the hardware converts).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHADERS = "/root/reference/backends/gpu-rt/shaders"
FILES = ["utils.glsl", "random.glsl", "structs.glsl", "disney.glsl", "intersection.glsl"]
OUT = os.path.join(HERE, "_ref")

FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")


def rewrite(text):
    text = re.sub(r'^\s*#include\s+"[^"]+"\s*$', "", text, flags=re.M)
    text = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = FLOAT_LIT.sub(lambda m: m.group(1) + "f", text)
    text = re.sub(r"\.xyz\b", ".xyz_()", text)
    text = re.sub(r"\bsign\(", "gl_sign_f(", text)
    text = re.sub(r"\buint\(", "to_uint(", text)
    text = re.sub(r"\bint\(", "to_int(", text)
    text = re.sub(r"\bfloat\(", "to_float(", text)
    return text


def available():
    return all(os.path.exists(os.path.join(SHADERS, f)) for f in FILES)


def build(verbose=False):
    """Returns the path of the library, or None where the reference checkout is absent."""
    if not available():
        return None
    import tempfile
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="glsl_ref_")   # the rewritten shader text is an intermediate: it exists only while the compiler runs
    gen = os.path.join(tmp, "glsl_ref_gen.inc")
    parts = []
    for f in FILES:
        parts.append(f"// ---- generated from {os.path.join(SHADERS, f)} by oracle/make_glsl_ref.py: do not commit\n")
        parts.append(rewrite(open(os.path.join(SHADERS, f)).read()))
    with open(gen, "w") as fh:
        fh.write("\n".join(parts))
    lib = os.path.join(OUT, "libglsl_ref.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           "-Wno-sign-compare", "-Wno-parentheses", "-I", HERE, "-I", tmp, "-o", lib, os.path.join(HERE, "glsl_ref_wrap.cpp")]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    if r.returncode != 0:
        raise RuntimeError("building the reference GLSL as C++ failed:\n" + r.stdout[-6000:])
    if verbose and r.stdout:
        print(r.stdout)
    return lib


if __name__ == "__main__":
    p = build(verbose=True)
    print(p if p else "reference checkout absent: nothing built")
    sys.exit(0)
