"""Time of cvx_handeye_solve per LM iteration at BASELINE config 2's sizes (development aid).
Usage: python scripts/handeye_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

for moving, n in ((True, 200), (False, 60), (False, 90), (False, 200)):
    G, T, xgt, x0 = cvx_amd.synth_handeye(n, 20260002, moving=moving)
    A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
    he = cvx_amd.HandEye(*cvx_amd.pairs_to_quat_trans(A, B))
    for iters in (5, 20):
        ho = capi.options("handeye")
        ho.function_tolerance = ho.gradient_tolerance = ho.parameter_tolerance = -1.0
        ho.max_iterations = iters
        he.solve(x0, ho)
        t = time.perf_counter()
        for _ in range(20):
            p, hs = he.solve(x0, ho)
        dt = (time.perf_counter() - t) / 20
        print("pairs %6d  %2d iterations: %.1f us per solve, %.2f us per LM iteration" % (len(A), hs.iterations, 1e6 * dt, 1e6 * dt / hs.iterations))
    ho = capi.options("handeye")        # the reference's own options: the solve as solveNonLinear runs it
    he.solve(x0, ho)
    t = time.perf_counter()
    for _ in range(20):
        p, hs = he.solve(x0, ho)
    dt = (time.perf_counter() - t) / 20
    print("pairs %6d  default options: %d iterations (%d accepted), %.1f us per solve" % (len(A), hs.iterations, hs.num_successful, 1e6 * dt))
    he.close()
