"""Where an LM iteration of the one-launch hand-eye solve (k_he_lm) spends its time, from in-kernel clock stamps of a
development build (-DHE_LM_PROF, see scripts/build_heprof.sh).
Usage: CVX_AMD_LIB=scripts/micro/libcvx_amd_heprof.so python scripts/handeye_prof.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

G, T, xgt, x0 = cvx_amd.synth_handeye(200, 20260002, moving=True)
A, B = cvx_amd.handeye_pairs(G, T, moving=True)
he = cvx_amd.HandEye(*cvx_amd.pairs_to_quat_trans(A, B))
ho = capi.options("handeye")
ho.function_tolerance = ho.gradient_tolerance = ho.parameter_tolerance = -1.0
ho.max_iterations = 20
for _ in range(3):
    p, s = he.solve(x0, ho)
    print("%d pairs, %d iterations: evaluation %.2f us, thread 0's scalar section %.2f us per iteration" % (
        len(A), s.iterations, s.trace_gradient[1000] / 100 / s.iterations, s.trace_gradient[1001] / 100 / s.iterations))
he.close()
