#!/bin/bash
# One rocprofv3 --pmc pass of the bench with the given SQ counters (at most 8), averaged per kernel.
# Usage: scripts/gpu_pmc_sq.sh <tag> <counter> [<counter> ...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/gpurun_out/sq_$tag -o sq -- python3 $root/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-configs > $root/gpurun_out/sq_$tag.json 2> $root/gpurun_out/sq_$tag.err || { tail -5 $root/gpurun_out/sq_$tag.err; exit 1; }
cd $root && python3 - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/sq_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("SQ_INSTS_VALU", 0))):
    n = len(cnt[k])
    print(k, n, {c: round(v / n) for c, v in sorted(acc[k].items())})
PY
