#!/bin/bash
# One rocprofv3 --pmc pass over a stand-alone binary (scripts/micro/...), counters averaged per kernel.
# Usage: scripts/gpu_pmc_bin.sh <tag> "<binary and its arguments>" <counter> [<counter> ...]   (at most 8 SQ counters)
tag=$1; cmd=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/gpurun_out/sqb_$tag -o sq -- $root/$cmd > $root/gpurun_out/sqb_$tag.out 2> $root/gpurun_out/sqb_$tag.err || { tail -5 $root/gpurun_out/sqb_$tag.err; exit 1; }
cd $root && python3 - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/sqb_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = len(cnt[k])
    print(k, n, {c: round(v / n) for c, v in sorted(acc[k].items())})
PY
