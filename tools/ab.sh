#!/bin/bash
# Same-box A/B of kernel variants, run on the GPU box:
#   make -C rfw-rs_amd/csrc variant VARIANT=x VFLAGS='-DRFW_SOMETHING=1'      (here; the .so travels with the snapshot)
#   gpurun -- 'LIBS="librfw_hip.so librfw_hip_x.so" bash tools/ab.sh'
# Three alternating runs per library of the headline workload (bench defaults), Mrays/s and ms per frame.
cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2 3; do for lib in $LIBS; do
  export RFW_HIP_LIB=$PWD/rfw-rs_amd/csrc/$lib
  timeout 200 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --procedural $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], {k: v['ms_sum_per_frame'] for k, v in d['roofline']['per_kernel'].items()})"
done; done
