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


def extract_tlas(out_path, source=None):
    """The fused TLAS build (k_tlas_fused of rfw-rs_amd/csrc/lbvh.hip) and the device functions it is made of."""
    src = open(source or os.path.join(ROOT, "rfw-rs_amd", "csrc", "lbvh.hip")).read()
    parts = [
        cut(src, "__device__ inline uint32_t f_order(float f)", "__global__ void k_init_bounds"),
        cut(src, "__device__ inline uint32_t expand10(uint32_t v)", "__global__ void k_morton"),
        cut(src, "// longest common prefix of keys i and j", "__global__ void k_hierarchy"),
        cut(src, "__device__ inline uint32_t node_slot(", "// bottom-up fit.  nbox[slot]"),
        cut(src, "// The same fit for ONE workgroup whose threads own several leaves each", "__global__ void k_flag_even_depth"),
        cut(src, "__device__ inline void emit4_node(", "__global__ void k_emit4"),
        cut(src, "__device__ inline DevBox instance_box(", "__global__ void k_instance_boxes"),
        cut(src, "constexpr uint32_t kFusedThreads = 1024;", "inline uint32_t blocks(uint32_t n)"),
    ]
    open(out_path, "w").write("\n".join(parts))


def whole_file(out_path, name, source=None):
    """A .hip file of the product as it stands, for a harness that compiles all of it against tests/emu/fake_hip; the one thing g++ cannot take
    is gfx950 assembly: the `s_waitcnt` statements (drain this wave's stores) become full fences."""
    src = open(source or os.path.join(ROOT, "rfw-rs_amd", "csrc", name)).read()
    src = src.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', "__atomic_thread_fence(__ATOMIC_SEQ_CST);")
    assert "asm volatile" not in src, "an assembly statement the emulation does not know"
    open(out_path, "w").write(src)


if __name__ == "__main__":
    extract(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
