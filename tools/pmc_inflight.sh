#!/bin/bash
# Counters of the DEFAULT bench mode (frames in flight: kernels of different frames share the CUs), per kernel family, summed over all launches:
#   gpurun -- 'bash tools/pmc_inflight.sh tag "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" ...'
# Every argument after the tag is one --pmc pass (counters in their own runs, never combined with traces).  tools/pmc_quick.sh is the
# one-frame-at-a-time twin (kernels alone on the chip).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pi_$TAG
mkdir -p $O
for kv in $ENVS; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-modes --procedural $BENCH_ARGS"
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$i -- $B > /dev/null 2> $O/pmc_$i.err || echo "pass $i ($pass) failed"
done
cd $R
python3 - "$O" "$TAG" <<'PY'
import sys, glob, csv, collections, json, os
O, tag = sys.argv[1:3]
res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfwhip::", "")
        res[n][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: {c: v for c, v in sorted(cs.items())} for k, cs in res.items() if k.startswith(("k_primary", "k_shade", "k_shadow", "k_extend", "k_assemble"))}
for k, c in out.items():
    if c.get("SQC_ICACHE_REQ"):
        c["icache_miss_rate"] = round(c.get("SQC_ICACHE_MISSES", 0) / c["SQC_ICACHE_REQ"], 4)
json.dump(out, open(os.path.join(os.path.dirname(O), f"{tag}_pmc_inflight.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_*
