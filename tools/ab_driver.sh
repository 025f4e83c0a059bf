#!/bin/bash
# A/B of the DRIVER's command (bench.py --steps 20 --warmup 5: 20 frames from an idle chip to an idle chip) next to the steady state
# (300 frames), per configuration of environment overrides:
#   gpurun -- 'CONFIGS="name:ENV=value,ENV=value name2:..." [REPS=3] bash tools/ab_driver.sh tag'
TAG=${1:-abd}
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/$TAG
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  (
    [ "$envs" != "$cfg" ] && for kv in ${envs//,/ }; do export "$kv"; done
    for rep in $(seq 1 ${REPS:-3}); do
      timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-modes $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/$TAG/${name}_drv_$rep.json
    done
    [ -z "$NO_STEADY" ] && timeout 300 python3 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline --no-modes $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/$TAG/${name}_steady.json
    python3 - "$name" gpurun_out/$TAG/${name}_ <<'PY'
import glob, json, sys
name, pre = sys.argv[1:3]
def val(f):
    try: return json.load(open(f))["value"]
    except Exception: return None
drv = [val(f) for f in sorted(glob.glob(pre + "drv_*.json"))]
print(f"{name:28s} driver {drv}  steady {val(pre + 'steady.json')}", flush=True)
PY
  )
done
