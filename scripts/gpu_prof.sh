#!/bin/bash
# rocprofv3 kernel trace of the bench; writes gpurun_out/prof_<tag>/ and a CSV summary.  Usage: scripts/gpu_prof.sh <tag>
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-configs > $root/gpurun_out/bench_prof_$tag.json 2> $root/gpurun_out/bench_prof_$tag.err || { tail -5 $root/gpurun_out/bench_prof_$tag.err; exit 1; }
cd $root && python3 scripts/rocpd_summary.py gpurun_out/prof_$tag/${tag}_results.db > gpurun_out/kernel_stats_$tag.csv && cut -c1-150 gpurun_out/kernel_stats_$tag.csv
