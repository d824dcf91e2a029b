#!/bin/bash
# A/B of two builds of the library on the box: bench.py's ms_per_step and per-phase times, alternating.
# Usage: [N=repetitions] scripts/ab_lib.sh <variant.so> ...
n=${N:-2}
for i in $(seq $n); do
  for lib in "" "$@"; do
    CVX_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); p=d['phases_ms_per_iteration']; print('lib=%-34s ms_per_step %.4f | K2 %.4f K3 %.4f (b2b %.4f) sums %.4f solve %.4f backsubst %.4f cost %.4f' % ('${lib:-shipped}', d['ms_per_step'], p['normal_eq'], p['schur'], d['roofline']['avg_launch_ms'], p['assemble'], p['solve'], p['backsubst'], p['cost']))" || exit 1
  done
done
