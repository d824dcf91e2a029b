"""Per-solve set-up cost of the calibration path at config 4 (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
o = capi.options("refine")
o.inner_iterations = 0      # the LM iteration itself
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0


def t(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return 1e3 * (time.perf_counter() - t0) / n


print("set_params            %.3f ms" % t(lambda: g.set_params(d["intr_init"], d["poses_init"], d["points_init"])))
for k in (0, 1, 2, 5, 10):
    o.max_iterations = k

    def run():
        g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
        g.solve(o)
    print("set_params + solve(%2d) %.3f ms" % (k, t(run)))
