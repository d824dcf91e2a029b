"""LM iteration time at BASELINE config 5 (100 000 views x 108 corners = 10.8 M observations) on one GPU
(development aid).  Usage: python scripts/config5_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

d = cvx_amd.synth_calib(8, 12500, 12, 9, 20260005)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
o = capi.options("refine")
o.inner_iterations = 0      # the LM iteration itself
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
o.max_iterations = 10
for rep in range(3):
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    t = time.perf_counter()
    s = g.solve(o)
    dt = time.perf_counter() - t
print("config5: N=%d V=%d ns=%d  %.3f ms per LM iteration (%d iterations), cost %.6g -> %.6g" % (
    d["N"], d["V"], g.ns, 1e3 * dt / s.iterations, s.iterations, s.initial_cost, s.final_cost))
g.close()
