"""Cuts the device source of the builder's workgroup phase out of rfw-rs_amd/csrc/sah_build.hip for tests/emu/k_small_emu.cpp (TEST INFRASTRUCTURE):
the types and helpers in front of the kernels, the 48-lane sweep, and phase 2 (k_small and what it calls) — the text as it ships, no edits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cut(text, start, end):
    a = text.index(start)
    return text[a:text.index(end, a)]


def extract(out_path, source=None):
    src = open(source or os.path.join(ROOT, "rfw-rs_amd", "csrc", "sah_build.hip")).read()
    parts = [
        cut(src, "constexpr int kBlock = 256;", "// ---------------------------------------------------------------- root"),
        cut(src, "struct SplitChoice {", "// one wavefront per active node;"),
        cut(src, "// ---------------------------------------------------------------- phase 2:", "// ---------------------------------------------------------------- BVH2 -> Node4"),
    ]
    open(out_path, "w").write("\n".join(parts))


if __name__ == "__main__":
    extract(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
