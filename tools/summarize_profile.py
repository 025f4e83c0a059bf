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
    if len(sys.argv) >= 7:
        # instruction mix: every SQ counter of the two extra passes, mean per launch, plus the derived figures DESIGN.md quotes
        counters = collections.defaultdict(lambda: collections.defaultdict(list))
        for d in sys.argv[5:7]:
            for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
                if "rfwhip" in r["Kernel_Name"]:
                    short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfwhip::", "")
                    counters[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        res = {}
        for k, c in sorted(counters.items()):
            m = {n: sum(v) / len(v) for n, v in c.items()}
            e = {n: int(v) for n, v in m.items()}
            valu, waves, gui = m.get("SQ_INSTS_VALU", 0.0), m.get("SQ_WAVES", 0.0), m.get("GRBM_GUI_ACTIVE", 0.0)
            if valu and waves:
                e["valu_insts_per_wave"] = round(valu / waves, 1)
            if valu and m.get("SQ_THREAD_CYCLES_VALU"):
                e["valu_lane_utilisation"] = round(m["SQ_THREAD_CYCLES_VALU"] / (m.get("SQ_ACTIVE_INST_VALU", valu) * 64.0), 3)
            if valu and gui:
                # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs.  Cycles between two vector instructions of a SIMD (an FP32 instruction
                # occupies it for 2: tools/probes/valu_issue_probe.hip)
                e["cycles_per_valu_inst_per_simd"] = round((gui / 8.0 * 1024.0) / max(valu, 1.0), 2)
            res[k] = e
        json.dump({"note": "means per launch from two rocprofv3 --pmc passes (instruction counts; VALU activity). cycles_per_valu_inst_per_simd = "
                           "kernel cycles * 1024 SIMDs / SQ_INSTS_VALU (an FP32 instruction occupies its SIMD for 2 cycles); valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)",
                   "kernels": res}, open(os.path.join(out, f"{tag}_pmc_valu.json"), "w"), indent=1)
    print("wrote profiles/", tag)


if __name__ == "__main__":
    main()
