#!/bin/bash
# Like tools/ab2.sh, but keeps the whole bench line per configuration and prints the traversal counters next to the rates:
#   gpurun -- 'CONFIGS="name:library:ENV=value,..." [BENCH_ARGS=...] bash tools/ab_json.sh tag'
TAG=${1:-abj}
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/$TAG
for rep in $(seq 1 ${REPS:-1}); do for cfg in $CONFIGS; do
  name=${cfg%%:*}; rest=${cfg#*:}; lib=${rest%%:*}; envs=${rest#*:}
  (
    export RFW_HIP_LIB=$PWD/rfw-rs_amd/csrc/$lib
    for kv in ${envs//,/ }; do export "$kv"; done
    timeout 400 python3 bench.py --steps ${STEPS:-300} --warmup ${WARMUP:-30} --no-cpu-baseline --procedural $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/$TAG/${name}_$rep.json
    python3 - gpurun_out/$TAG/${name}_$rep.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]; c = r.get("contract", {}).get("per_kernel", {})
print(sys.argv[2], d["value"], d["ms_per_step"], {k[:28]: v["Mrays_per_s"] for k, v in d["config"]["modes"].items() if not v.get("is_value")}, {k: v["ms"] for k, v in c.items()},
      "nodes/ray", r.get("nodes_per_ray"), "tris/ray", r.get("tris_per_ray"), "packets", {k: v for k, v in (r.get("primary_packets") or {}).items() if "per_wavefront" in k},
      "bvh", {k: d["config"]["bvh"].get(k) for k in ("blas_nodes",)}, "sync_warm", (d["config"]["bvh"].get("build") or {}).get("synchronize_warm_ms"), flush=True)
PY
  )
done; done
