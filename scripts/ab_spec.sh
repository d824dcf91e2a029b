#!/bin/bash
# A/B of the speculated steps on the box: bench.py's ms_per_step with the device-side decisions and with every decision
# waiting for the host (the default), alternating.  Usage: scripts/ab_spec.sh [repetitions]
n=${1:-3}
for i in $(seq $n); do
  for hl in 0 1; do
    CVX_CALIB_SPECULATE=$((1-hl)) timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('host_loop=$hl ms_per_step %.4f full_refine_ms %.3f' % (d['ms_per_step'], d['full_refine_to_convergence']['ms_total']))" || exit 1
  done
done
