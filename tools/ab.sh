#!/bin/bash
# Same-box A/B of kernel variants, run on the GPU box:
#   make -C rfw-rs_amd/csrc variant VARIANT=x VFLAGS='-DRFW_SOMETHING=1'      (here; the .so travels with the snapshot)
#   gpurun -- 'LIBS="librfw_hip.so librfw_hip_x.so" bash tools/ab.sh'
# Three alternating runs per library of the headline workload (bench defaults: one render() per frame, 8 frame slots, 16 views):
# Mrays/s, ms per frame, the one-frame-at-a-time mode and its per-kernel HIP-event times.
cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2 3; do for lib in $LIBS; do
  export RFW_HIP_LIB=$PWD/rfw-rs_amd/csrc/$lib
  timeout 300 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --procedural $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline'].get('contract',{}).get('per_kernel',{})
print('$lib', d['value'], d['ms_per_step'], {k: v['Mrays_per_s'] for k, v in d['config']['modes'].items() if not v.get('is_value')}, {k: v['ms'] for k, v in c.items()}, d['roofline'].get('lane_utilisation',{}).get('shadow'))"
done; done
