#!/usr/bin/env python3
"""Per-basic-block table of ONE kernel of a gfx950 ISA listing:  python3 tools/isa_blocks.py [--build] <kernel substring> [listing.s]

tools/isa_stats.py gives a kernel's totals; this prints its basic blocks in listing order — label, vector instructions (and how many of them are
`v_mov`: register copies), scalar instructions, memory instructions, and where its branches go (^ backward, v forward) — which is how the
hot trip of a loop is read: follow the backward branches, add up the blocks on the way.  Round 5 found with it that 24 of the 128 vector
instructions of a trip of k_shadow were copies and that a third of a trip was scalar mask arithmetic.
The kernel is chosen by a substring of its DEMANGLED name, e.g. "k_shadow<false, false>" or "k_extend_stream<false>".
`--loops` prints one line per kernel instead: instructions inside loops (vector / scalar / v_mov), for every trace kernel."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_stats  # noqa: E402


def kernels_of(lines):
    """[(first line, mangled, demangled)] of every kernel body in the listing."""
    heads = [(i, ln.split(":")[0]) for i, ln in enumerate(lines) if re.match(r"^_Z\w+:", ln)]
    names = isa_stats.demangle([m for _, m in heads])
    return [(i, m, d) for (i, m), d in zip(heads, names)]


def body_of(lines, start):
    out = []
    for ln in lines[start + 1:]:
        out.append(ln)
        if ln.strip().startswith("s_endpgm"):
            break
    return out


def blocks_of(body):
    blocks, cur = [], ["entry", [], False]
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), [], "Loop" in ln]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur[1].append(t.split(";")[0].strip())
    blocks.append(cur)
    return blocks


def count(instrs):
    v = sum(1 for x in instrs if x.startswith("v_"))
    mv = sum(1 for x in instrs if x.startswith("v_mov"))
    s = sum(1 for x in instrs if x.startswith("s_"))
    mem = sum(1 for x in instrs if re.match(r"(global_|buffer_|flat_|ds_|scratch_)", x))
    return v, mv, s, mem


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--build" in sys.argv:
        path, _ = isa_stats.build()
    else:
        path = args[1] if len(args) > 1 else os.path.join(os.environ.get("TMPDIR", "/tmp"), "rfw_isa", "kernels.s")
    lines = open(path).read().splitlines()
    ks = kernels_of(lines)
    if "--loops" in sys.argv:
        for i, _, d in ks:
            if not re.search(r"k_(shadow|extend|primary|query)", d):
                continue
            inl = [x for b in blocks_of(body_of(lines, i)) if b[2] for x in b[1]]
            v, mv, s, _ = count(inl)
            print(f"{d:46s} in loops: vector {v:5d} (v_mov {mv:4d})  scalar {s:5d}")
        return
    if not args:
        raise SystemExit(__doc__)
    hits = [(i, m, d) for i, m, d in ks if args[0] in d]
    if not hits:
        raise SystemExit(f"no kernel whose demangled name contains {args[0]!r}; e.g. {[d for _, _, d in ks][:5]}")
    i, _, d = hits[0]
    blocks = blocks_of(body_of(lines, i))
    index = {b[0]: n for n, b in enumerate(blocks)}
    print(f"# {d}: {len(blocks)} blocks   (no. label  vector (v_mov)  scalar  memory  | branches: ^ backward, v forward)")
    for n, (label, instrs, in_loop) in enumerate(blocks):
        v, mv, s, mem = count(instrs)
        to = []
        for x in instrs:
            if "branch" in x:
                t = x.split()[-1]
                to.append(("^" if index.get(t, 1 << 30) <= n else "v") + t.replace(".LBB", "") + "(" + x.split()[0][2:].replace("cbranch_", "") + ")")
        print(f"{n:3d} {label.replace('.LBB', ''):9s} {'L' if in_loop else ' '} v={v:3d} (mov {mv:2d}) s={s:3d} m={mem:2d}  {' '.join(to)}")


if __name__ == "__main__":
    main()
