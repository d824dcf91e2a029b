#!/bin/bash
# Everything profiles/rNN* holds for one build, in one call on the box: the default bench line (CPU baselines included),
# the rocprofv3 kernel trace of the bench, the HBM traffic passes, the fp64 instruction counters, two SQ counter passes,
# and the kernel trace of an inner-iteration pass.  Each step under its own time limit; the first failure ends the call.
# Usage: scripts/gpu_round_profile.sh <tag>
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root && mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/bench_full_$tag.json 2> gpurun_out/bench_full_$tag.err || { tail -5 gpurun_out/bench_full_$tag.err; exit 1; }
echo "bench done"
bash scripts/gpu_prof.sh $tag > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
echo "kernel trace done"
bash scripts/gpu_pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
echo "traffic done"
bash scripts/gpu_pmc_fp64.sh $tag > gpurun_out/fp64_$tag.log 2>&1 || { tail -5 gpurun_out/fp64_$tag.log; exit 1; }
echo "fp64 counters done"
bash scripts/gpu_pmc_sq.sh ${tag}_a GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES > gpurun_out/sq_${tag}_a.txt 2>&1 || { tail -5 gpurun_out/sq_${tag}_a.txt; exit 1; }
bash scripts/gpu_pmc_sq.sh ${tag}_b SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > gpurun_out/sq_${tag}_b.txt 2>&1 || { tail -5 gpurun_out/sq_${tag}_b.txt; exit 1; }
echo "SQ counters done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_inner_$tag -o inner -- python3 $root/scripts/inner_time.py > $root/gpurun_out/inner_$tag.out 2> $root/gpurun_out/inner_$tag.err || { tail -5 $root/gpurun_out/inner_$tag.err; exit 1; }
cd $root && python3 scripts/rocpd_summary.py gpurun_out/prof_inner_$tag/inner_results.db > gpurun_out/inner_kernel_stats_$tag.csv && cat gpurun_out/inner_$tag.out
echo "inner pass trace done"
