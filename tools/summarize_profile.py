#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/**.csv) into the small summaries committed under profiles/.

usage: tools/summarize_profile.py <round-tag> <kernel-trace-dir> [<pmc-fetch-dir> <pmc-write-dir> [<pmc-inst-dir> <pmc-valu-dir> [<more pmc dirs> ...]]]
Writes
  profiles/<tag>_kernel_stats_one_at_a_time.csv  copy of rocprofv3 --stats of the trace directory given (tools/measure.sh: one frame at a time)
  profiles/<tag>_pmc.json          mean FETCH_SIZE / WRITE_SIZE per kernel (KB as rocprofv3 reports them) + bytes per launch with the gfx950
                                   correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane loads => x2
  profiles/<tag>_pmc_valu.json     instruction mix and VALU activity per launch
  profiles/<tag>_pmc_cache.json    every counter of the remaining passes (vector L1 = TCP, L2 = TCC, texture-address unit = TA), mean per
                                   launch, plus derived hit rates and request bytes.  A pass whose counters the profiler rejected is skipped."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(d, pat, required=True):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        if required:
            raise SystemExit(f"no {pat} under {d}")
        return None
    return f[0]


def short_name(k):
    return k.split("(")[0].replace("void ", "").replace("rfwhip::", "")


def pmc_means(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def all_counters(dirs):
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = find(d, "*_counter_collection.csv", required=False) if os.path.isdir(d) else None
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if "rfwhip" in r["Kernel_Name"]:
                counters[short_name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return counters


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(find(kt, "*_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats_one_at_a_time.csv"))
    sys.path.insert(0, os.path.dirname(out))
    from bench import kernel_source_hash
    stamp = {"source_hash": kernel_source_hash(), "source_hash_note": "sha1 of rfw-rs_amd/csrc sources at measurement time (bench.kernel_source_hash): bench.py ignores this file when its checkout differs"}
    if len(sys.argv) >= 5:
        fetch, nf = pmc_means(sys.argv[3], "FETCH_SIZE")
        write, _ = pmc_means(sys.argv[4], "WRITE_SIZE")
        res = {}
        for k in sorted(set(fetch) | set(write)):
            if "rfwhip" not in k:
                continue
            f, w = fetch.get(k, 0.0), write.get(k, 0.0)
            res[short_name(k)] = {"launches_sampled": nf.get(k, 0), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                                  "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
        json.dump({**stamp, "note": "means per launch; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half of a "
                           "16-B-per-lane stream; uncalibrated for divergent gathers, so treat as an upper bound there).  These are the L2's "
                           "fabric-side requests: Infinity-Cache hits are included (MI355X_MICROARCH.md §HBM), so true HBM bytes are lower still",
                   "kernels": res}, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    if len(sys.argv) >= 7:
        # instruction mix: every SQ counter of the two extra passes, mean per launch, plus the derived figures DESIGN.md quotes
        counters = all_counters(sys.argv[5:7])
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
                # occupies it for 2: tools/probes/valu_issue_probe.hip).  With frames in flight the kernels of different frames share the
                # chip, so this per-launch figure then overstates what ONE kernel would need alone
                e["cycles_per_valu_inst_per_simd"] = round((gui / 8.0 * 1024.0) / max(valu, 1.0), 2)
            res[k] = e
        json.dump({**stamp, "note": "means per launch from two rocprofv3 --pmc passes (instruction counts; VALU activity). cycles_per_valu_inst_per_simd = "
                           "kernel cycles * 1024 SIMDs / SQ_INSTS_VALU (an FP32 instruction occupies its SIMD for 2 cycles); valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)",
                   "kernels": res}, open(os.path.join(out, f"{tag}_pmc_valu.json"), "w"), indent=1)
    if len(sys.argv) >= 8:
        counters = all_counters(sys.argv[7:])
        res = {}
        for k, c in sorted(counters.items()):
            m = {n: sum(v) / len(v) for n, v in c.items()}
            e = {n: int(v) for n, v in m.items()}
            hit, miss = m.get("TCC_HIT_sum"), m.get("TCC_MISS_sum")
            if hit is not None and miss is not None and hit + miss > 0:
                e["l2_hit_rate"] = round(hit / (hit + miss), 4)
                e["l2_request_bytes_per_launch"] = int((hit + miss) * 128)  # one request = one 128-B line
            acc, tcc_rd = m.get("TCP_TOTAL_CACHE_ACCESSES_sum"), m.get("TCP_TCC_READ_REQ_sum")
            if acc and tcc_rd is not None:
                e["l1_hit_rate"] = round(1.0 - tcc_rd / acc, 4)  # share of the L1's line accesses that did not become an L2 read
            res[k] = e
        json.dump({**stamp, "note": "means per launch, one rocprofv3 --pmc pass per pair of counters.  l1_hit_rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES; "
                           "l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md §L2); l2_request_bytes = (TCC_HIT + TCC_MISS) x 128 B",
                   "kernels": res}, open(os.path.join(out, f"{tag}_pmc_cache.json"), "w"), indent=1)
    print("wrote profiles/", tag)


if __name__ == "__main__":
    main()
