#!/bin/bash
# rocprofv3 kernel-trace of scripts/k2_time.py for each ablation library given; prints the k_normal_eq<128,0> average
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export CVX_AMD_LIB=$root/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/abl_$tag -o $tag -- python3 $root/scripts/k2_time.py > $root/gpurun_out/abl_$tag.out 2> $root/gpurun_out/abl_$tag.err || { tail -5 $root/gpurun_out/abl_$tag.err; exit 1; }
  (cd $root && python3 scripts/rocpd_summary.py gpurun_out/abl_$tag/${tag}_results.db | grep -E "k_normal_eq<128, 0|k_backsubst" | cut -d, -f2-6 | sed "s/^/$tag: /")
done
