#!/bin/bash
# FP64 utilisation counters of the bench kernels (one rocprofv3 --pmc pass, 7 SQ + 1 GRBM counters).
# Usage: scripts/gpu_pmc_fp64.sh <tag>
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_fp64 -o fp64 -- python3 $root/bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-configs > $root/gpurun_out/pmc_${tag}_fp64.json 2> $root/gpurun_out/pmc_${tag}_fp64.err || { tail -5 $root/gpurun_out/pmc_${tag}_fp64.err; exit 1; }
cd $root && python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_${tag}_fp64/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
names = ["SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
with open("gpurun_out/fp64_counters_${tag}.csv", "w") as o:
    o.write("Kernel,Launches," + ",".join(n + "_per_launch" for n in names) + "\n")
    for k in sorted(tot, key=lambda k: -tot[k]["GRBM_GUI_ACTIVE"]):
        n = max(cnt[k], 1)
        o.write('"%s",%d,' % (k, n) + ",".join("%.0f" % (tot[k][c] / n) for c in names) + "\n")
print(open("gpurun_out/fp64_counters_${tag}.csv").read())
PY
