#!/usr/bin/env python3
"""Per-kernel resource table of a gfx950 ISA listing: python3 tools/isa_stats.py [--build] [listing.s] [filter]

--build compiles rfw-rs_amd/csrc/kernels.hip to ISA with the Makefile's flags (device side only, a few seconds, no GPU needed) into
$TMPDIR/rfw_isa/kernels.s and reads that.  For every kernel (demangled, `rfwhip::` stripped): VGPRs, SGPRs, scratch bytes
(.amdhsa_private_segment_fixed_size: anything > 0 is a spill or a private array), LDS bytes, and the static instruction counts of its
body by issue class — v_* (vector ALU), s_* (scalar), vector memory, LDS, scratch_* — plus the number of scratch loads that sit inside a
loop (between a label and a backward branch to it).  Used by tests/test_isa_budget.py (no trace kernel may touch scratch) and by hand to
diff two builds."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rfw-rs_amd", "csrc")


def makefile_flags():
    txt = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    m = re.search(r"^FLAGS\s*=\s*(.*)$", txt, re.M)
    return m.group(1).replace("$(ARCH)", "gfx950").split()


def build(source="kernels.hip", extra=()):
    out_dir = os.path.join(tempfile.gettempdir(), "rfw_isa")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, os.path.splitext(source)[0] + ".s")
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + list(extra) + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, source)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return out, r.stderr


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.splitlines()
        if len(out) == len(names):
            return [re.sub(r"\(.*$", "", o.replace("void ", "").replace("rfwhip::", "")) for o in out]
    except Exception:
        pass
    return names


def parse(path):
    kernels = collections.OrderedDict()
    cur, body = None, []
    lines = open(path).read().splitlines()
    bodies = {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith("."):
            cur = m.group(1)
            bodies[cur] = []
            continue
        if cur is not None:
            if ln.strip().startswith(".section") or ln.strip().startswith(".end_amdhsa_kernel"):
                pass
            bodies[cur].append(ln)
            if ln.strip() == "s_endpgm" or ln.strip().startswith("s_endpgm"):
                pass
    meta = {}
    cur = None
    for ln in lines:
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            cur = m.group(1)
            meta[cur] = {}
            continue
        if cur:
            m = re.match(r"^\s*\.amdhsa_(\w+)\s+(\S+)", ln)
            if m:
                meta[cur][m.group(1)] = m.group(2)
            if ".end_amdhsa_kernel" in ln:
                cur = None
    names = [k for k in bodies if k in meta]
    pretty = demangle(names)
    for k, p in zip(names, pretty):
        ins = collections.Counter()
        labels = {}
        scratch_load_lines = []
        text = []
        for ln in bodies[k]:
            s = ln.strip()
            if s.startswith(".amdhsa_kernel") or s.startswith(".section") or s.startswith(".rodata"):
                break
            text.append(s)
        for i, s in enumerate(text):
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, s in enumerate(text):
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", s)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        for i, s in enumerate(text):
            if not s or s.startswith((";", ".", "/")) or s.endswith(":"):
                continue
            op = s.split()[0]
            if op.startswith("scratch_"):
                ins["scratch"] += 1
                if op.startswith("scratch_load") and any(a <= i <= b for a, b in loops):
                    ins["scratch_loads_in_loops"] += 1
            elif op.startswith(("global_", "flat_", "buffer_")):
                ins["vmem"] += 1
            elif op.startswith("ds_"):
                ins["lds"] += 1
            elif op.startswith("v_"):
                ins["valu"] += 1
                if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                    ins["lane_ops"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                ins["smem"] += 1
            elif op.startswith("s_"):
                ins["salu"] += 1
        md = meta[k]
        kernels[p if p not in kernels else p + "'"] = {
            "vgpr": int(md.get("next_free_vgpr", 0)), "sgpr": int(md.get("next_free_sgpr", 0)),
            "scratch_bytes": int(md.get("private_segment_fixed_size", 0)), "lds_bytes": int(md.get("group_segment_fixed_size", 0)),
            **{c: ins.get(c, 0) for c in ("valu", "salu", "smem", "vmem", "lds", "scratch", "scratch_loads_in_loops", "lane_ops")},
        }
    return kernels


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = None
    if "--build" in sys.argv:
        path, _ = build()
    if args and args[0].endswith(".s"):
        path = args.pop(0)
    if not path:
        path = os.path.join(tempfile.gettempdir(), "rfw_isa", "kernels.s")
    flt = args[0] if args else ""
    ks = parse(path)
    cols = ("vgpr", "sgpr", "scratch_bytes", "lds_bytes", "valu", "salu", "smem", "vmem", "lds", "scratch", "scratch_loads_in_loops", "lane_ops")
    print("kernel".ljust(44) + " ".join(c[:10].rjust(10) for c in cols))
    for name, k in ks.items():
        if flt in name:
            print(name[:43].ljust(44) + " ".join(str(k[c]).rjust(10) for c in cols))


if __name__ == "__main__":
    main()
