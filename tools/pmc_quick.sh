#!/bin/bash
# Quick counter look at ONE kernel family under one set of environment variables, run on the GPU box:
#   gpurun -- 'ENVS="RFW_PACKET_TRACE=1" KERNEL=k_primary bash tools/pmc_quick.sh tagname "SQ_INSTS_VALU SQ_INSTS_SALU ..." "SQC_DCACHE_REQ ..." ...'
# Every further argument is one --pmc pass (counters in their own runs, never combined with traces).  Prints the mean per launch of every
# counter for the kernels whose name contains $KERNEL, and leaves gpurun_out/<tag>_pmc_quick.json.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pq_$TAG
mkdir -p $O
for kv in $ENVS; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-modes --procedural --frames-in-flight 1 $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/bench.json 2> $O/trace.err
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$i -- $B > /dev/null 2> $O/pmc_$i.err || echo "pass $i ($pass) failed"
done
cd $R
python3 - "$O" "${KERNEL:-k_}" "$TAG" <<'PY'
import sys, glob, csv, collections, json, os
O, kern, tag = sys.argv[1:4]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            res[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfwhip::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} for k, cs in res.items()}
for f in glob.glob(os.path.join(O, "trace", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ", "").replace("rfwhip::", "")
        if kern in n:
            out.setdefault(n, {})["avg_ns"] = float(r["AverageNs"]); out[n]["calls"] = int(r["Calls"])
json.dump(out, open(os.path.join(os.path.dirname(O), f"{tag}_pmc_quick.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/trace $O/pmc_*
