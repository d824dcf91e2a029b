"""LM iteration time of the calibration path at BASELINE configs 1, 3, 4 (development aid).
Usage: python scripts/config_times.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

for name, (C_, VP, bw, bh, seed) in {"config1": (1, 20, 9, 6, 20260001), "config3": (2, 500, 12, 9, 20260003),
                                     "config4": (8, 1250, 12, 9, 20260004)}.items():
    d = cvx_amd.synth_calib(C_, VP, bw, bh, seed)
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh)
    o = capi.options("refine")
    o.inner_iterations = 0      # the LM iteration itself
    o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
    o.max_iterations = 10
    for rep in range(3):
        g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
        t = time.perf_counter()
        s = g.solve(o)
        dt = time.perf_counter() - t
    print("%s: N=%d V=%d ns=%d  %.3f ms per LM iteration (%d iterations)" % (name, d["N"], d["V"], g.ns, 1e3 * dt / s.iterations, s.iterations))
    g.close()
