#!/bin/bash
# A/B of any bench.py command line per configuration of environment overrides (one line per run: value, ms per frame, per-frame synchronize):
#   gpurun -- 'CONFIGS="name:ENV=value,ENV=value name2:X=0" [REPS=2] BENCH_ARGS="--workload spheres10k --steps 200 --warmup 30" bash tools/ab_bench.sh tag'
TAG=${1:-abb}
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/$TAG
for rep in $(seq 1 ${REPS:-2}); do for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  (
    [ "$envs" != "$cfg" ] && for kv in ${envs//,/ }; do export "$kv"; done
    timeout 600 python3 bench.py --gpus 1 --no-cpu-baseline --no-modes $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/$TAG/${name}_$rep.json
    python3 - "$name" gpurun_out/$TAG/${name}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"{sys.argv[1]:24s} {d['value']:9.1f} Mrays/s  {d['ms_per_step']:.4f} ms/frame  sync {d['config'].get('per_frame_synchronize_ms')} render-call {d['config'].get('per_frame_render_call_ms')}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
  )
done; done
