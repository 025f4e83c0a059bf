#!/bin/bash
# Same-box A/B of kernel variants AND run-time options, run on the GPU box:
#   gpurun -- 'CONFIGS="base:librfw_hip.so: pk1:librfw_hip.so:RFW_PACKET_TRACE=1 pk8:librfw_hip_pw8.so:RFW_PACKET_TRACE=1" bash tools/ab2.sh'
# A configuration is name:library:ENV=value[,ENV=value...].  REPS (default 3) alternating runs of the bench defaults (one render() per frame,
# 8 frame slots, 16 views; BENCH_ARGS adds flags): Mrays/s, ms per frame, the other modes, per-kernel HIP-event ms, lane utilisation.
cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in $(seq 1 ${REPS:-3}); do for cfg in $CONFIGS; do
  name=${cfg%%:*}; rest=${cfg#*:}; lib=${rest%%:*}; envs=${rest#*:}
  (
    export RFW_HIP_LIB=$PWD/rfw-rs_amd/csrc/$lib
    for kv in ${envs//,/ }; do export "$kv"; done
    timeout 300 python3 bench.py --steps ${STEPS:-400} --warmup 40 --no-cpu-baseline --procedural $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline'].get('contract',{}).get('per_kernel',{})
print('$name', d['value'], d['ms_per_step'], {k: v['Mrays_per_s'] for k, v in d['config']['modes'].items() if not v.get('is_value')}, {k: v['ms'] for k, v in c.items()}, d['roofline'].get('lane_utilisation',{}), d['config'].get('timed_frame_equals_oracle'))"
  )
done; done
