#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/**.csv) into the small summaries committed under profiles/.

usage: tools/summarize_profile.py <round-tag> <kernel-trace-dir> [<pmc-fetch-dir> <pmc-write-dir>]
Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --stats) and profiles/<tag>_pmc.json (mean FETCH_SIZE /
WRITE_SIZE per kernel, KB as rocprofv3 reports them, plus bytes per launch with the gfx950 correction of
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane loads => x2)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        raise SystemExit(f"no {pat} under {d}")
    return f[0]


def pmc_means(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(find(kt, "*_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        fetch, nf = pmc_means(sys.argv[3], "FETCH_SIZE")
        write, _ = pmc_means(sys.argv[4], "WRITE_SIZE")
        res = {}
        for k in sorted(set(fetch) | set(write)):
            if "rfwhip" not in k:
                continue
            short = k.split("(")[0].replace("void ", "").replace("rfwhip::", "")
            f, w = fetch.get(k, 0.0), write.get(k, 0.0)
            res[short] = {"launches_sampled": nf.get(k, 0), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                          "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
        json.dump({"note": "means per launch; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half of a "
                           "16-B-per-lane stream; uncalibrated for divergent gathers, so treat as an upper bound there)",
                   "kernels": res}, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    print("wrote profiles/", tag)


if __name__ == "__main__":
    main()
