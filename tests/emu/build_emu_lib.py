"""TEST INFRASTRUCTURE: librfw_hip as a CPU library — every source file of rfw-rs_amd/csrc compiled for the HOST (the ROCm clang as an x86 compiler:
it knows the vector extensions the kernels use) against tests/emu/fake_hip, its kernels running under wave_emu.h.  Same C ABI, so the Python
bindings (RFW_HIP_LIB=<this library>) and with them the parity tests drive the kernels' SOURCE on the CPU.  Built into a directory the caller
names, never into the package: nothing of the product loads it, and the product without its HIP library still fails loudly.
Substitutions (the only edits to the text): gfx950 assembly — the s_waitcnt drain becomes a fence, the packet traversal's scalar add-with-carry
its C statement, register-class constraints of empty asm barriers become "r", the issue probe's instruction mixes go — and the binding of
llvm.amdgcn.writelane by name becomes a function."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "rfw-rs_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
UNITS = ["kernels.hip", "lbvh.hip", "sah_build.hip", "api_scene.cpp", "api_frame.cpp", "api_exchange.cpp", "api_query.cpp", "bvh_host.cpp"]


def patched(text):
    text = text.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', "__atomic_thread_fence(__ATOMIC_SEQ_CST);")
    text = text.replace('asm("s_cmp_lg_u64 %1, 0\\n\\ts_addc_u32 %0, %0, 0" : "+s"(sp) : "s"(m[i]) : "scc");', "sp += m[i] != 0ull ? 1 : 0;")
    text = text.replace('extern "C" __device__ int rfw_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");',
                        "inline int rfw_llvm_writelane(int value, int lane, int old) { return (int)emu::lane() == lane ? value : old; }")
    text = text.replace('#include "../../include/', '#include "')
    text = re.sub(r'asm volatile\(""\s*:\s*"\+[vs]"', 'asm volatile("" : "+r"', text)
    text = re.sub(r'asm volatile\("[vs]_[^;]*;', "/* gfx950 assembly */;", text)   # (the issue probe: never launched here)
    return text


def build(out_dir, jobs=4, replace=None, extra_flags=()):
    """replace: {file name in csrc: path of another text for it} — an experiment's version of a source file (experiments/*/...patch applied)"""
    os.makedirs(out_dir, exist_ok=True)
    for name in os.listdir(CSRC):
        if name.endswith((".hip", ".cpp", ".h", ".inc")):
            src = (replace or {}).get(name, os.path.join(CSRC, name))
            open(os.path.join(out_dir, name), "w").write(patched(open(src).read()))
    flags = ["-std=c++20", "-O1", "-g0", "-fPIC", "-ffp-contract=off", "-pthread", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-Wno-unknown-attributes", "-Wno-ignored-attributes",
             "-Wno-unused-value", "-D__forceinline__=inline", "-D__host__=", "-D__device__=", "-D__global__=", "-DRFW_EMULATED=1", "-x", "c++", "-I", out_dir, "-I", os.path.join(EMU, "fake_hip"), "-I", EMU,
             "-I", os.path.join(ROOT, "include")] + list(extra_flags)
    procs, objs = [], []
    for u in UNITS:
        obj = os.path.join(out_dir, u.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        procs.append((u, subprocess.Popen([CLANG] + flags + ["-c", os.path.join(out_dir, u), "-o", obj], stderr=subprocess.PIPE, text=True)))
        if len(procs) >= jobs:
            u0, p0 = procs.pop(0)
            err = p0.communicate()[1]
            if p0.returncode != 0:
                raise RuntimeError(u0 + "\n" + err[-6000:])
    for u0, p0 in procs:
        err = p0.communicate()[1]
        if p0.returncode != 0:
            raise RuntimeError(u0 + "\n" + err[-6000:])
    lib = os.path.join(out_dir, "librfw_hip_emu.so")
    r = subprocess.run([CLANG, "-shared", "-fPIC", "-pthread", "-o", lib] + objs + [f for f in extra_flags if f.startswith(("-fsanitize", "-shared-libsan"))], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-6000:])
    return lib


if __name__ == "__main__":
    print(build(sys.argv[1], replace=dict(a.split("=", 1) for a in sys.argv[2:] if not a.startswith("-")), extra_flags=[a for a in sys.argv[2:] if a.startswith("-")]))
