#!/bin/bash
# The 1 / 2 / 4 / 8 GPU table of SURVEY §8(e), ready to run on an 8 x MI355X node:
#     bash tools/scale.sh [tag] [steps]          (one node, from the repository root; needs the built libraries: python3 -c 'import __graft_entry__ as g; g.build()')
# Runs the plain command `python3 bench.py --gpus N` (bench.py starts its N ranks itself, one per GPU, when no launcher has: launch_ranks;
# under `python -m torch.distributed.run` the same script is one of the launcher's ranks) for every N in $GPUS and, at N > 1, for every
# (collective, gather format) pair in $VARIANTS; prints one row per run with the whole-job rate, the speed-up over N = 1, and whether the
# sharded frame equalled the single-GPU frame (config.sharded_frame_equals_single_gpu_frame: rank 0 re-renders the last timed frames on one
# GPU and compares in the format that travelled).  Raw lines: gpurun_out/<tag>/scale_*.json; table: gpurun_out/<tag>/scale.md.
# On a 1-GPU box only the N = 1 row runs (RCCL refuses two ranks on one device); RFW_BENCH_DIST_BACKEND=gloo lets 2 ranks share a GPU for a
# functional check of the other rows (tests/test_gpu_parity.py does that).
# `bash tools/scale.sh --dry` is the rehearsal for a box WITHOUT the GPUs: every (exchange, format) pair that can run with two ranks on ONE
# device (torch's collective through the gloo hook, the peer-store exchange through real IPC handles — also with the fall-back kind of flag
# memory forced, RFW_P2P_FLAGS_FINEGRAINED=1) on a small frame, and it FAILS LOUDLY (exit 1, the failing pairs named) unless every run exits 0
# with config.sharded_frame_equals_single_gpu_frame true.  `native` (RCCL inside the library) cannot be rehearsed this way — RCCL refuses two
# ranks on one device: everything AROUND its ncclAllGather call runs with three ranks of one process through the library's loop-back hub
# (tests/test_gpu_parity.py::test_library_exchange_through_the_loopback_hub), the call itself with a one-rank communicator (tests/test_gpu_api.py).
if [ "$1" = "--dry" ]; then
  R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/scale_dry; mkdir -p "$OUT"; cd "$R" || exit 1
  export HSA_ENABLE_IPC_MODE_LEGACY=0 RFW_BENCH_DIST_BACKEND=gloo
  PORT=29710; FAILED=""
  for v in torch:bgra8 torch:f16 torch:f32 p2p:bgra8 p2p:f16 p2p:f32 p2p:bgra8:finegrained-flags p2p:f32:cached-data torch:bgra8:batched p2p:bgra8:batched; do
    coll=${v%%:*}; rest=${v#*:}; fmt=${rest%%:*}; extra=""
    case "$v" in *finegrained-flags) extra="RFW_P2P_FLAGS_FINEGRAINED=1";; *cached-data) extra="RFW_P2P_DATA_CACHED=1";; esac
    barg=""; case "$v" in *batched) barg="--batch 8";; esac   # (default: one exchange per FRAME, north_star's protocol; batched: one per 8 frames)
    PORT=$((PORT + 1)); f=$OUT/dry_${v//:/_}.json
    env $extra timeout 600 python3 bench.py --gpus 2 \
      --steps 12 --warmup 4 --workload cornell --width 320 --height 200 --no-cpu-baseline --collective "$coll" --gather-format "$fmt" $barg > "$f" 2> "${f%.json}.err"
    rc=$?
    ok=$(python3 -c "import json,sys; l=[x for x in open('$f') if x.startswith('{')]; d=json.loads(l[-1]) if l else {}; print(int(d.get('config',{}).get('sharded_frame_equals_single_gpu_frame') is True and d.get('n_gpus')==2 and d.get('config',{}).get('ranks_seen')==2))" 2>/dev/null)
    if [ "$rc" -ne 0 ] || [ "$ok" != "1" ]; then FAILED="$FAILED $v(rc=$rc)"; echo "FAIL $v"; tail -5 "${f%.json}.err"; else echo "ok   $v"; fi
  done
  if [ -n "$FAILED" ]; then echo "scale.sh --dry: FAILED:$FAILED"; exit 1; fi
  echo "scale.sh --dry: every pair ran and reproduced the single-GPU frame"; exit 0
fi
TAG=${1:-scale}
STEPS=${2:-240}
GPUS=${GPUS:-"1 2 4 8"}
VARIANTS=${VARIANTS:-"torch:bgra8 torch:f16 torch:f32 native:bgra8 native:f32 p2p:bgra8 p2p:f16 p2p:f32"}
# every row above exchanges the framebuffer once per FRAME (north_star's protocol, bench.py's default).  The same with 8 frames traced per launch
# and ONE exchange per batch (bench.py --batch 8, the rfw_hip_render_batch extension):
BATCH_VARIANTS=${BATCH_VARIANTS:-"native:bgra8 p2p:bgra8"}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
PORT=29610
ROWS=$OUT/rows.txt
: > "$ROWS"
run() { # n collective format [batched]
  local n=$1 coll=$2 fmt=$3 lat=$4 f=$OUT/scale_n${1}_${2}_${3}${4:+_batched}.json
  local extra=""
  [ -n "$lat" ] && extra="--batch 8"   # 8 frames per launch, one exchange per batch
  if [ "$n" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps "$STEPS" --warmup 24 --no-cpu-baseline --no-modes > "$f" 2> "${f%.json}.err"
  else
    python3 bench.py --gpus "$n" \
      --steps "$STEPS" --warmup 24 --no-cpu-baseline --collective "$coll" --gather-format "$fmt" $extra > "$f" 2> "${f%.json}.err"
  fi
  echo "$n $coll ${fmt}${lat:+(one-exchange-per-8-frames)} $f" >> "$ROWS"
}
for n in $GPUS; do
  if [ "$n" -gt "$HAVE" ] && [ -z "$RFW_BENCH_DIST_BACKEND" ]; then echo "skipping N = $n: this box has $HAVE GPU(s)"; continue; fi
  if [ "$n" -eq 1 ]; then run 1 - -; continue; fi
  for v in $VARIANTS; do run "$n" "${v%%:*}" "${v##*:}"; done
  for v in $BATCH_VARIANTS; do run "$n" "${v%%:*}" "${v##*:}" batched; done
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
