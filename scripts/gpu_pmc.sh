#!/bin/bash
# HBM traffic counters of the bench kernels, one rocprofv3 --pmc pass per counter (FETCH_SIZE takes 3 of the
# 4 TCC slots, WRITE_SIZE 2: they cannot share a pass).  Usage: scripts/gpu_pmc.sh <tag>
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$ctr -o $ctr -- python3 $root/bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-configs > $root/gpurun_out/pmc_${tag}_$ctr.json 2> $root/gpurun_out/pmc_${tag}_$ctr.err || { tail -5 $root/gpurun_out/pmc_${tag}_$ctr.err; exit 1; }
done
cd $root && ls gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE && python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > gpurun_out/traffic_$tag.json && cat gpurun_out/traffic_$tag.json
