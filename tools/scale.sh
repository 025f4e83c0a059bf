#!/bin/bash
# The 1 / 2 / 4 / 8 GPU table of SURVEY §8(e), ready to run on an 8 x MI355X node:
#     bash tools/scale.sh [tag] [steps]          (one node, from the repository root; needs the built libraries: python3 -c 'import __graft_entry__ as g; g.build()')
# Runs bench.py exactly as the driver does (python -m torch.distributed.run, one rank per GPU) for every N in $GPUS and, at N > 1, for every
# (collective, gather format) pair in $VARIANTS; prints one row per run with the whole-job rate, the speed-up over N = 1, and whether the
# sharded frame equalled the single-GPU frame (config.sharded_frame_equals_single_gpu_frame: rank 0 re-renders the last timed frames on one
# GPU and compares in the format that travelled).  Raw lines: gpurun_out/<tag>/scale_*.json; table: gpurun_out/<tag>/scale.md.
# On a 1-GPU box only the N = 1 row runs (RCCL refuses two ranks on one device); RFW_BENCH_DIST_BACKEND=gloo lets 2 ranks share a GPU for a
# functional check of the other rows (tests/test_gpu_parity.py does that).
TAG=${1:-scale}
STEPS=${2:-240}
GPUS=${GPUS:-"1 2 4 8"}
VARIANTS=${VARIANTS:-"torch:bgra8 torch:f16 torch:f32 native:bgra8 p2p:bgra8 p2p:f32"}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
PORT=29610
ROWS=$OUT/rows.txt
: > "$ROWS"
run() { # n collective format
  local n=$1 coll=$2 fmt=$3 f=$OUT/scale_n${1}_${2}_${3}.json
  if [ "$n" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps "$STEPS" --warmup 24 --no-cpu-baseline --no-modes > "$f" 2> "${f%.json}.err"
  else
    PORT=$((PORT + 1))
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$n" \
      --steps "$STEPS" --warmup 24 --no-cpu-baseline --collective "$coll" --gather-format "$fmt" > "$f" 2> "${f%.json}.err"
  fi
  echo "$n $coll $fmt $f" >> "$ROWS"
}
for n in $GPUS; do
  if [ "$n" -gt "$HAVE" ] && [ -z "$RFW_BENCH_DIST_BACKEND" ]; then echo "skipping N = $n: this box has $HAVE GPU(s)"; continue; fi
  if [ "$n" -eq 1 ]; then run 1 - -; continue; fi
  for v in $VARIANTS; do run "$n" "${v%%:*}" "${v##*:}"; done
done
python3 - "$ROWS" > "$OUT/scale.md" <<'PY'
import json, sys
rows, base = [], None
for line in open(sys.argv[1]):
    n, coll, fmt, path = line.split()
    try:
        out = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        rows.append((int(n), coll, fmt, None)); continue
    if int(n) == 1:
        base = out["value"]
    rows.append((int(n), coll, fmt, out))
print("| GPUs | exchange | tiles travel as | Mrays/s (whole job) | ms per frame | speed-up | efficiency | sharded frame = single-GPU frame |")
print("|---|---|---|---|---|---|---|---|")
for n, coll, fmt, out in rows:
    if out is None:
        print(f"| {n} | {coll} | {fmt} | failed (see the .err file) | | | | |"); continue
    sp = out["value"] / base if base else float("nan")
    print(f"| {n} | {coll} | {fmt} | {out['value']:.0f} | {out['ms_per_step']:.4f} | {sp:.2f} | {sp / n:.2f} | {out['config'].get('sharded_frame_equals_single_gpu_frame')} |")
PY
cat "$OUT/scale.md"
