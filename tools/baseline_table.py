#!/usr/bin/env python3
"""Prints the result rows of BASELINE.md §4 from the committed bench lines of a round:  python3 tools/baseline_table.py r06
(`HBM` columns = the line's roofline.frac: HBM-side bytes by the counters / duration / 8 TB/s; `binding` = the unit the kernel sits closest to)"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
names = {"c2": "C2 262 k tris, primary+shadow, static", "c3": "C3 C2 + 10 k animated instances (TLAS rebuilt on device every frame)",
         "c4": "C4 geometry (1.05 M tris), primary+shadow = the headline metric", "c4path": "C4 path tracer, max path length 3, NEE",
         "c32m": "control: the same atrium at 33.6 M triangles (far outside every cache), primary+shadow"}
print("| Config | Mrays/s (`value`) | ms/frame | rays/frame | other modes (Mrays/s) | dominant kernel alone: issue / TA / L2 / **HBM** | timed region: issue / TA / L2 / **HBM** | CPU restatement Mrays/s (threads) | timed frames vs oracle |")
print("|---|---|---|---|---|---|---|---|---|")
for cfg in ("c2", "c3", "c4", "c4path", "c32m"):
    if not os.path.exists(os.path.join(root, f"{tag}_{cfg}_bench.json")):
        continue
    d = json.load(open(os.path.join(root, f"{tag}_{cfg}_bench.json")))
    r = d["roofline"]
    modes = "; ".join(f"{k}: {v['Mrays_per_s']:.0f}" for k, v in d["config"].get("modes", {}).items() if not v.get("is_value") and "Mrays_per_s" in v and not k.startswith("BASELINE config"))
    fr = lambda c: " / ".join(f"{c[k]['frac']:.2f}" if k in c else "-" for k in ("valu_issue", "l1_ta", "l2", "hbm"))
    cb = d.get("cpu_baseline") or {}
    eq = d["config"].get("timed_frame_equals_oracle")
    print(f"| {names[cfg]} | **{d['value']:.0f}** ({d['config']['mode']}) | {d['ms_per_step']:.3f} | {d['config']['rays_per_frame'] / 1e6:.2f} M | {modes} | "
          f"`{r['kernel']}` {r['contract']['avg_launch_ms']:.3f} ms: {fr(r['ceilings'])} | {fr(r['timed_region']['ceilings'])} | "
          f"{(str(cb.get('value')) + ' (' + str(cb.get('cores')) + ')') if cb else '-'} | {'bit-identical (first and last)' if eq and eq.get('all') else ('-' if not eq else str(eq))} |")
