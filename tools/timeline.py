#!/usr/bin/env python3
"""Condenses a `rocprofv3 --kernel-trace --output-format csv` run of `bench.py --steps K --warmup W --no-modes` into the timeline of its TIMED
region: one row per kernel (start, end, duration in ms from the first timed kernel; kernel; queue), and a summary line — how long the K
frames took on the GPU, and how long the chip ran fewer than 4 kernels at once at the start and at the end (the fill and the drain).
    python3 tools/timeline.py <dir with *_kernel_trace.csv> K W [out.csv]
The timed frames are found through their `k_primary*` launches: the first primaries that overlap one another are the warm-up's, the next K
the timed region's (the counting passes before and the probes after run one frame at a time).  DESIGN.md §10 reads such a timeline."""
import csv
import glob
import os
import re
import sys


def short(n):
    m = re.search(r"k_\w+", n)
    return m.group(0) if m else n[:28]


def main():
    d, K, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    path = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[-1]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(path)))
    prim = [k for k in ks if k[2].startswith("k_primary")]
    # the warm-up and the timed frames are the first primaries that OVERLAP one another (the counting passes before them and the probes after
    # them run one frame at a time): the first overlap is the first warm-up frame
    first = next((i for i in range(len(prim) - 1) if prim[i + 1][0] < prim[i][1]), None)
    if first is None or first + W + K > len(prim):
        raise SystemExit("no overlapping run of primaries found")
    t0 = prim[first + W][0]
    t_next = prim[first + W + K][0] if first + W + K < len(prim) else ks[-1][1] + 1
    sel = [k for k in ks if t0 <= k[0] < t_next and k[2].startswith(("k_", "__amd"))]
    # cut at the first gap of more than 0.3 ms with nothing running (whatever bench.py runs next)
    end = sel[0][1]
    keep = []
    for k in sel:
        if k[0] > end + 300000:
            break
        keep.append(k)
        end = max(end, k[1])
    rows = [((k[0] - t0) / 1e6, (k[1] - t0) / 1e6, k[2], k[3]) for k in keep]
    total = max(r[1] for r in rows)
    ev = sorted([(r[0], 1) for r in rows if r[2].startswith("k_")] + [(r[1], -1) for r in rows if r[2].startswith("k_")])
    n, last, thin = 0, 0.0, 0.0
    resident = {}  # kernels running at once -> ms
    for t, dlt in ev:
        if n < 4:
            thin += t - last
        resident[n] = resident.get(n, 0.0) + (t - last)
        n += dlt
        last = t
    out = sys.argv[4] if len(sys.argv) > 4 else None
    if out:
        with open(out, "w") as f:
            f.write("start_ms,end_ms,duration_ms,kernel,queue\n")
            for r in rows:
                f.write(f"{r[0]:.3f},{r[1]:.3f},{r[1] - r[0]:.3f},{r[2]},{r[3]}\n")
    by = {}
    for r in rows:
        by.setdefault(r[2], []).append(r[1] - r[0])
    import json
    print(json.dumps({"frames": K, "gpu_ms": round(total, 3), "ms_per_frame": round(total / K, 4), "ms_with_fewer_than_4_kernels_running": round(thin, 3),
                      "share_of_time_with_at_least_3_kernels_running": round(sum(v for k, v in resident.items() if k >= 3) / max(total, 1e-9), 3),
                      "ms_by_kernels_running_at_once": {str(k): round(v, 3) for k, v in sorted(resident.items())},
                      "resident_ms_mean_max": {k: [round(sum(v) / len(v), 3), round(max(v), 3)] for k, v in by.items()}}, indent=1))


if __name__ == "__main__":
    main()
